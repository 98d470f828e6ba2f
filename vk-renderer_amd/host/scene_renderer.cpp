#include "scene_renderer.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

// scene_renderer.cpp:8-44: formats, usages and the depth mip count floor(log2(max(w,h))) + 1.
Gbuffer::Gbuffer(rendergraph::RenderGraph &graph, uint32_t width, uint32_t height) : w {width}, h {height} {
  const auto tiling = VK_IMAGE_TILING_OPTIMAL;
  const auto color_usage = VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  const auto depth_usage = VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT|VK_IMAGE_USAGE_SAMPLED_BIT|VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  const uint32_t depth_mips = uint32_t(std::floor(std::log2(std::max(width, height)))) + 1;

  auto color = [&](VkFormat fmt, uint32_t cw, uint32_t ch) {
    return graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo {fmt, VK_IMAGE_ASPECT_COLOR_BIT, cw, ch}, tiling, color_usage);
  };
  albedo = color(VK_FORMAT_R8G8B8A8_SRGB, width, height);
  normal = color(VK_FORMAT_R16G16_UNORM, width, height);
  velocity_vectors = color(VK_FORMAT_R16G16_SFLOAT, width, height);
  downsampled_normals = color(VK_FORMAT_R16G16_UNORM, width/2, height/2);
  downsampled_velocity_vectors = color(VK_FORMAT_R16G16_SFLOAT, width/2, height/2);
  material = color(VK_FORMAT_R8G8B8A8_SRGB, width, height);

  gpu::ImageInfo depth_info {VK_FORMAT_D24_UNORM_S8_UINT, VK_IMAGE_ASPECT_DEPTH_BIT|VK_IMAGE_ASPECT_STENCIL_BIT, width, height, 1, depth_mips, 1};
  depth = graph.create_image(VK_IMAGE_TYPE_2D, depth_info, tiling, depth_usage);
  prev_depth = graph.create_image(VK_IMAGE_TYPE_2D, depth_info, tiling, depth_usage|VK_IMAGE_USAGE_TRANSFER_DST_BIT);
  frame_hiz = depth;
  frame_normals = downsampled_normals;
  frame_albedo = albedo;
}

void Gbuffer::enable_tiling(rendergraph::RenderGraph &graph, uint32_t full_width, uint32_t full_height) {
  tiled = true;
  const uint32_t frame_mips = uint32_t(std::floor(std::log2(std::max(full_width, full_height)))) + 1;
  gpu::ImageInfo hiz {VK_FORMAT_D24_UNORM_S8_UINT, VK_IMAGE_ASPECT_DEPTH_BIT, full_width/2, full_height/2, 1, frame_mips - 1, 1};
  frame_hiz = graph.create_frame_image(hiz);
  frame_normals = graph.create_frame_image(gpu::ImageInfo {VK_FORMAT_R16G16_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, full_width/2, full_height/2});
  frame_albedo = graph.create_frame_image(gpu::ImageInfo {VK_FORMAT_R8G8B8A8_SRGB, VK_IMAGE_ASPECT_COLOR_BIT, full_width, full_height});
}

// ---- scene ------------------------------------------------------------------------------------------------
namespace scene {

CompiledScene make_scene(const Vertex *vertices, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                         const FlatDraw *draws, uint32_t draw_count, const TextureData *textures, uint32_t texture_count)
{
  CompiledScene out;
  // scene.cpp:285-296: one vertex and one index buffer for the whole file
  auto upload = [](const void *src, uint64_t bytes, VkBufferUsageFlags usage) {
    auto buf = gpu::create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, std::max<uint64_t>(bytes, 4), usage);
    if (bytes) std::memcpy(buf->get_mapped_ptr(), src, bytes);
    return buf;
  };
  out.vertex_buffer = upload(vertices, sizeof(Vertex) * uint64_t(vertex_count), VK_BUFFER_USAGE_TRANSFER_DST_BIT);
  out.index_buffer = upload(indices, sizeof(uint32_t) * uint64_t(index_count), VK_BUFFER_USAGE_TRANSFER_DST_BIT);
  // images.cpp:32-49: every texture is an RGBA8_SRGB image with a full mip chain
  auto sampler_info = gpu::DEFAULT_SAMPLER;
  sampler_info.addressModeU = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  sampler_info.addressModeV = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  out.samplers.push_back(gpu::create_sampler(sampler_info));
  for (uint32_t i = 0; i < texture_count; i++) {
    const TextureData &t = textures[i];
    gpu::ImageInfo info {VK_FORMAT_R8G8B8A8_SRGB, VK_IMAGE_ASPECT_COLOR_BIT, t.width, t.height, 1, t.mip_levels, 1};
    auto img = std::make_shared<gpu::Image>(info, gpu::FrameWindow {});
    for (uint32_t m = 0; m < t.mip_levels; m++) img->upload_mip(m, t.levels[m]);
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

}

// ---- SceneRenderer (scene_renderer.cpp:46-220) -----------------------------------------------------------------
void SceneRenderer::init_pipeline(rendergraph::RenderGraph &graph, const Gbuffer &buffer) {
  owner = &graph;
  gpu::Registers regs {};
  regs.depth_stencil.depthTestEnable = VK_TRUE;
  regs.depth_stencil.depthWriteEnable = VK_TRUE;

  opaque_taa_pipeline = gpu::create_graphics_pipeline();
  opaque_taa_pipeline.set_program("gbuf_opaque_taa");
  opaque_taa_pipeline.set_registers(regs);
  opaque_taa_pipeline.set_vertex_input({});
  opaque_taa_pipeline.set_rendersubpass({true, {VK_FORMAT_R8G8B8A8_SRGB, VK_FORMAT_R16G16_UNORM, VK_FORMAT_R8G8B8A8_SRGB, VK_FORMAT_R16G16_SFLOAT}});

  auto sampler_info = gpu::DEFAULT_SAMPLER;
  sampler_info.addressModeU = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  sampler_info.addressModeV = VK_SAMPLER_ADDRESS_MODE_REPEAT;
  sampler = gpu::create_sampler(sampler_info);

  // host-visible here: the program premultiplies view_projection * model per draw on the host
  transform_buffer = graph.create_buffer(VMA_MEMORY_USAGE_CPU_TO_GPU, sizeof(glm::mat4) * 1000, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT|VK_BUFFER_USAGE_TRANSFER_DST_BIT);

  scene_textures.reserve(target.textures.size());
  for (auto tex_desc : target.textures) {
    auto &img = target.images[tex_desc.image_index];
    gpu::ImageViewRange range {VK_IMAGE_VIEW_TYPE_2D, VK_IMAGE_ASPECT_COLOR_BIT, 0, img->get_mip_levels(), 0, 1};
    texture_views.emplace_back(new gpu::ImageViewObject {img.get(), range});
    scene_textures.push_back({(VkImageView)texture_views.back().get(), target.samplers[tex_desc.sampler_index]});
  }
  bindless_textures = gpu::allocate_descriptor_set(opaque_taa_pipeline.get_layout(1), {std::max<uint32_t>(1, uint32_t(scene_textures.size()))});
  if (scene_textures.size())
    gpu::write_set(bindless_textures, gpu::ArrayOfImagesBinding {0, scene_textures});
  (void)buffer;
}

static void node_process(const scene::BaseNode &node, std::vector<SceneRenderer::DrawCall> &draw_calls, std::vector<glm::mat4> &transforms, const glm::mat4 &acc) {
  const auto transform = acc * node.transform;
  const uint32_t transform_id = uint32_t(transforms.size()/2);
  if (node.mesh_index >= 0) {
    transforms.push_back(transform);
    transforms.push_back(glm::transpose(glm::inverse(transform)));
    draw_calls.push_back(SceneRenderer::DrawCall {transform_id, uint32_t(node.mesh_index)});
  }
  for (auto &child : node.children)
    node_process(child, draw_calls, transforms, transform);
}

void SceneRenderer::update_scene() {
  if (!owner) throw std::runtime_error {"SceneRenderer::update_scene before init_pipeline"};
  std::vector<glm::mat4> transforms;
  draw_calls.clear();
  for (auto &node : target.base_nodes)
    node_process(node, draw_calls, transforms, glm::mat4 {1.f});
  auto &buf = owner->get_buffer(transform_buffer);
  if (sizeof(glm::mat4) * transforms.size() > buf->get_size()) throw std::runtime_error {"Too many scene transforms"};
  std::memcpy(buf->get_mapped_ptr(), transforms.data(), sizeof(glm::mat4) * transforms.size());
}

void SceneRenderer::draw_taa(rendergraph::RenderGraph &graph, const Gbuffer &gbuffer, const DrawTAAParams &params) {
  struct Data { rendergraph::ImageViewId albedo, normal, material, depth, velocity; };
  struct PushData { uint32_t transform_index, albedo_index, mr_index, flags; };
  static_assert(sizeof(vkr_gbuf_const) == 2 * sizeof(glm::mat4) + 2 * sizeof(glm::vec4), "GbufConst must match the C-ABI");
  vkr_gbuf_const consts {};
  std::memcpy(&consts.view_projection, &params.mvp, sizeof(glm::mat4));
  std::memcpy(&consts.prev_view_projection, &params.prev_mvp, sizeof(glm::mat4));
  std::memcpy(consts.jitter, &params.jitter, sizeof(glm::vec4));
  std::memcpy(consts.fovy_aspect_znear_zfar, &params.fovy_aspect_znear_zfar, sizeof(glm::vec4));

  graph.add_task<Data>("GbufferPass",
    [&](Data &in, rendergraph::RenderGraphBuilder &builder) {
      in.albedo = builder.use_color_attachment(gbuffer.albedo, 0, 0);
      in.normal = builder.use_color_attachment(gbuffer.normal, 0, 0);
      in.material = builder.use_color_attachment(gbuffer.material, 0, 0);
      in.depth = builder.use_depth_attachment(gbuffer.depth, 0, 0);
      in.velocity = builder.use_color_attachment(gbuffer.velocity_vectors, 0, 0);
      builder.use_storage_buffer(transform_buffer, VK_SHADER_STAGE_VERTEX_BIT);
    },
    [=](Data &in, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      cmd.set_framebuffer(gbuffer.w, gbuffer.h, {
        resources.get_image_range(in.albedo), resources.get_image_range(in.normal), resources.get_image_range(in.material),
        resources.get_image_range(in.velocity), resources.get_image_range(in.depth)});

      cmd.bind_pipeline(opaque_taa_pipeline);
      cmd.clear_color_attachments(0.f, 0.f, 0.f, 0.f);
      cmd.clear_depth_attachment(1.f);
      cmd.bind_viewport(0.f, 0.f, float(gbuffer.w), float(gbuffer.h), 0.f, 1.f);
      cmd.bind_scissors(0, 0, gbuffer.w, gbuffer.h);
      cmd.bind_vertex_buffers(0, {target.vertex_buffer->api_buffer()}, {0ul});
      cmd.bind_index_buffer(target.index_buffer->api_buffer(), 0, VK_INDEX_TYPE_UINT32);

      auto blk = cmd.allocate_ubo<vkr_gbuf_const>();
      *blk.ptr = consts;
      auto set = resources.allocate_set(opaque_taa_pipeline, 0);
      gpu::write_set(set,
        gpu::UBOBinding {0, cmd.get_ubo_pool(), blk},
        gpu::SSBOBinding {1, resources.get_buffer(transform_buffer)});
      cmd.bind_descriptors_graphics(0, {set}, {blk.offset});
      cmd.bind_descriptors_graphics(1, {bindless_textures}, {});

      for (const auto &draw_call : draw_calls) {
        for (const auto &prim : target.root_meshes[draw_call.mesh].primitives) {
          const auto &material = target.materials[prim.material_index];
          PushData pc {};
          pc.transform_index = draw_call.transform;
          pc.albedo_index = (material.albedo_tex_index < scene_textures.size())? material.albedo_tex_index : scene::INVALID_TEXTURE;
          pc.mr_index = (material.metalic_roughness_index < scene_textures.size())? material.metalic_roughness_index : scene::INVALID_TEXTURE;
          pc.flags = material.clip_alpha? 0xff : 0;
          cmd.push_constants_graphics(VK_SHADER_STAGE_VERTEX_BIT|VK_SHADER_STAGE_FRAGMENT_BIT, 0, sizeof(PushData), &pc);
          cmd.draw_indexed(prim.index_count, 1, prim.index_offset, int32_t(prim.vertex_offset), 0);
        }
      }
      cmd.end_renderpass();
    });
}
