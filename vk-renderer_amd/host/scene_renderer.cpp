#include "scene_renderer.hpp"

#include <algorithm>
#include <cmath>

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
