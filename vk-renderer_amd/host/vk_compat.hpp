// vk_compat.hpp — the handful of Vulkan / VMA names the reference's pass code spells out
// (formats, usage / aspect / stage flags, handles).  When the real <vulkan/vulkan.h> is on
// the include path it is used instead; the numeric values below are the Vulkan 1.2 ones so
// code compiled either way agrees.  Nothing here talks to a Vulkan driver: the handles are
// opaque pointers into the HIP-backed gpu:: layer.
#ifndef VKR_VK_COMPAT_HPP_INCLUDED
#define VKR_VK_COMPAT_HPP_INCLUDED
#include <cstdint>

#if __has_include(<vulkan/vulkan.h>) && !defined(VKR_FORCE_VK_COMPAT)
#include <vulkan/vulkan.h>
#else
typedef uint32_t VkFlags;
typedef VkFlags VkImageAspectFlags, VkImageUsageFlags, VkShaderStageFlags, VkBufferUsageFlags;
typedef uint32_t VkBool32;
typedef uint64_t VkDeviceSize;
#define VK_TRUE 1u
#define VK_FALSE 0u
enum VkFormat {
  VK_FORMAT_UNDEFINED = 0, VK_FORMAT_R8_UNORM = 9, VK_FORMAT_R8G8B8A8_UNORM = 37, VK_FORMAT_R8G8B8A8_SRGB = 43,
  VK_FORMAT_R16_SFLOAT = 76, VK_FORMAT_R16G16_UNORM = 77, VK_FORMAT_R16G16_SFLOAT = 83,
  VK_FORMAT_R16G16B16A16_UNORM = 91, VK_FORMAT_R16G16B16A16_SFLOAT = 97, VK_FORMAT_R32_UINT = 98, VK_FORMAT_R32_SFLOAT = 100,
  VK_FORMAT_R32G32B32A32_SFLOAT = 109, VK_FORMAT_D24_UNORM_S8_UINT = 129
};
enum { VK_IMAGE_ASPECT_COLOR_BIT = 1, VK_IMAGE_ASPECT_DEPTH_BIT = 2, VK_IMAGE_ASPECT_STENCIL_BIT = 4 };
enum {
  VK_IMAGE_USAGE_TRANSFER_SRC_BIT = 0x1, VK_IMAGE_USAGE_TRANSFER_DST_BIT = 0x2, VK_IMAGE_USAGE_SAMPLED_BIT = 0x4,
  VK_IMAGE_USAGE_STORAGE_BIT = 0x8, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT = 0x10,
  VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT = 0x20
};
enum {
  VK_BUFFER_USAGE_TRANSFER_DST_BIT = 0x2, VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT = 0x10,
  VK_BUFFER_USAGE_STORAGE_BUFFER_BIT = 0x20, VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT = 0x100
};
enum { VK_SHADER_STAGE_VERTEX_BIT = 0x1, VK_SHADER_STAGE_FRAGMENT_BIT = 0x10, VK_SHADER_STAGE_COMPUTE_BIT = 0x20 };
enum VkImageType { VK_IMAGE_TYPE_1D = 0, VK_IMAGE_TYPE_2D = 1, VK_IMAGE_TYPE_3D = 2 };
enum VkImageTiling { VK_IMAGE_TILING_OPTIMAL = 0, VK_IMAGE_TILING_LINEAR = 1 };
enum VkImageViewType { VK_IMAGE_VIEW_TYPE_2D = 1, VK_IMAGE_VIEW_TYPE_2D_ARRAY = 5 };
enum VkCompareOp { VK_COMPARE_OP_NEVER = 0, VK_COMPARE_OP_LESS = 1, VK_COMPARE_OP_ALWAYS = 7 };
enum VkFilter { VK_FILTER_NEAREST = 0, VK_FILTER_LINEAR = 1 };
enum VkSamplerMipmapMode { VK_SAMPLER_MIPMAP_MODE_NEAREST = 0, VK_SAMPLER_MIPMAP_MODE_LINEAR = 1 };
enum VkSamplerAddressMode { VK_SAMPLER_ADDRESS_MODE_REPEAT = 0, VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE = 2, VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_BORDER = 3 };
struct VkExtent2D { uint32_t width, height; };
struct VkExtent3D { uint32_t width, height, depth; };
struct VkSamplerCreateInfo {
  VkFilter magFilter, minFilter;
  VkSamplerMipmapMode mipmapMode;
  VkSamplerAddressMode addressModeU, addressModeV, addressModeW;
  float minLod, maxLod;
};
enum VkIndexType { VK_INDEX_TYPE_UINT16 = 0, VK_INDEX_TYPE_UINT32 = 1 };
struct VkDispatchIndirectCommand { uint32_t x, y, z; };
typedef struct VkBuffer_T* VkBuffer;
typedef struct VkSampler_T* VkSampler;
typedef struct VkImageView_T* VkImageView;
typedef struct VkDescriptorSet_T* VkDescriptorSet;
typedef struct VkDescriptorSetLayout_T* VkDescriptorSetLayout;
typedef struct VkRenderPass_T* VkRenderPass;
typedef struct VkCommandBuffer_T* VkCommandBuffer;
typedef struct VkAccelerationStructureKHR_T* VkAccelerationStructureKHR;  // named by GTAO::add_main_rt_pass only
#endif

#if __has_include(<vk_mem_alloc.h>) && !defined(VKR_FORCE_VK_COMPAT)
#include <vk_mem_alloc.h>
#else
enum VmaMemoryUsage { VMA_MEMORY_USAGE_UNKNOWN = 0, VMA_MEMORY_USAGE_GPU_ONLY = 1, VMA_MEMORY_USAGE_CPU_ONLY = 2, VMA_MEMORY_USAGE_CPU_TO_GPU = 3 };
#endif
#endif
