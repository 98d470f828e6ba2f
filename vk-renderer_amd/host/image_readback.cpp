// image_readback.cpp — see image_readback.hpp.
#include "image_readback.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <vector>

namespace {
void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error {std::string {what} + ": " + hipGetErrorString(e)};
}
}

ReadBackID ReadBackSystem::read_image(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId image) {
  return read_image(graph, image, 0, 0, 0);
}

ReadBackID ReadBackSystem::read_image(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId image, VkImageAspectFlags, uint32_t mip, uint32_t layer) {
  struct Nothing {};
  const auto &desc = graph.get_descriptor(image);
  const uint32_t w = std::max(1u, desc.width >> mip), h = std::max(1u, desc.height >> mip);
  const uint32_t texel = vkr_format_bytes(gpu::to_vkr_format(desc.format));  // throws "Unsupported ..." for foreign formats
  if (mip >= desc.mip_levels) throw std::runtime_error {"ReadBack: mip outside the image"};

  void *host = nullptr;
  hip_check(hipHostMalloc(&host, size_t(w) * h * texel, hipHostMallocDefault), "ReadBack: pinned allocation failed");
  std::shared_ptr<void> pinned {host, [](void *p) { (void)hipHostFree(p); }};

  const ReadBackID id = next_request_id++;
  requests[id] = Request {graph.get_frames_count() + 1, w, h, desc.format, texel, pinned};

  graph.add_task<Nothing>("ImageRead",
    [&](Nothing &, rendergraph::RenderGraphBuilder &builder) {
      builder.transfer_read(image, mip, 1, layer, 1);
    },
    [=](Nothing &, rendergraph::RenderResources &resources, gpu::CmdContext &cmd) {
      const auto &img = resources.get_image(image);
      vkr_img view = layer == 0 ? img->describe(mip, 1) : img->describe_layer(layer);
      // the view is the window held by this process: rows of `w` texels, pitch from the descriptor
      hip_check(hipMemcpy2DAsync(pinned.get(), size_t(w) * texel, view.base, view.pitch_bytes[0], size_t(w) * texel, h,
                                 hipMemcpyDeviceToHost, (hipStream_t)cmd.get_stream()),
                "ReadBack: copy failed");
    });
  return id;
}

void ReadBackSystem::after_submit(rendergraph::RenderGraph &graph) {
  bool synced = false;
  for (auto it = requests.begin(); it != requests.end();) {
    Request &r = it->second;
    if (r.wait_frames > 0) {
      r.wait_frames--;
      ++it;
      continue;
    }
    if (!synced) {
      hip_check(hipStreamSynchronize((hipStream_t)graph.get_stream()), "ReadBack: stream synchronisation failed");
      synced = true;
    }
    ReadBackData out {};
    out.width = r.width;
    out.height = r.height;
    out.texel_size = r.texel_size;
    out.texel_fmt = r.texel_fmt;
    const size_t n = size_t(r.width) * r.height * r.texel_size;
    out.bytes.reset(new uint8_t[n]);
    std::memcpy(out.bytes.get(), r.pinned.get(), n);
    processed_requests[it->first] = std::move(out);
    it = requests.erase(it);
  }
}

ReadBackData ReadBackSystem::get_data(ReadBackID id) {
  auto node = processed_requests.extract(id);
  if (node.empty()) throw std::out_of_range {"ReadBack: no data for this request"};
  return std::move(node.mapped());
}

void ReadBackSystem::clear() {
  requests.clear();
  processed_requests.clear();
}

// ---- capture writers ----------------------------------------------------------------------------
void write_depth_csv(const ReadBackData &image, const std::string &path) {
  const uint32_t *words = reinterpret_cast<const uint32_t *>(image.bytes.get());
  std::ofstream file {path, std::ios::trunc};
  if (!file) throw std::runtime_error {"write_depth_csv: cannot open " + path};
  file << "y, ";
  for (uint32_t x = 0; x < image.width; x++) file << x << (x + 1 != image.width ? "," : "");
  file << "\n";
  for (uint32_t y = 0; y < image.height; y++) {
    file << y << "," << std::hex;
    for (uint32_t x = 0; x < image.width; x++)
      file << "0x" << (0xffffffu & words[size_t(y) * image.width + x]) << (x + 1 != image.width ? "," : "");
    file << std::dec << "\n";
  }
}

namespace {
struct Crc32 {
  uint32_t table[256];
  Crc32() {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[n] = c;
    }
  }
  uint32_t run(uint32_t crc, const uint8_t *p, size_t n) const {
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
  }
};
void put_u32(std::vector<uint8_t> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back(uint8_t(x >> s)); }
void put_chunk(std::ofstream &f, const Crc32 &crc, const char tag[4], const std::vector<uint8_t> &data) {
  std::vector<uint8_t> head;
  put_u32(head, (uint32_t)data.size());
  f.write((const char *)head.data(), 4);
  f.write(tag, 4);
  f.write((const char *)data.data(), (std::streamsize)data.size());
  uint32_t c = crc.run(0xFFFFFFFFu, (const uint8_t *)tag, 4);
  c = crc.run(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
  std::vector<uint8_t> tail;
  put_u32(tail, c);
  f.write((const char *)tail.data(), 4);
}
}

bool write_png_rgba8(const std::string &path, uint32_t width, uint32_t height, const uint8_t *rgba) {
  std::ofstream f {path, std::ios::binary | std::ios::trunc};
  if (!f) return false;
  static const Crc32 crc;
  const uint8_t sig[8] {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  f.write((const char *)sig, 8);
  std::vector<uint8_t> ihdr;
  put_u32(ihdr, width);
  put_u32(ihdr, height);
  for (uint8_t b : {uint8_t(8), uint8_t(6), uint8_t(0), uint8_t(0), uint8_t(0)}) ihdr.push_back(b);  // 8-bit RGBA, no interlace
  put_chunk(f, crc, "IHDR", ihdr);
  // raw scanlines: filter byte 0 + row
  const size_t row = size_t(width) * 4;
  std::vector<uint8_t> raw;
  raw.reserve((row + 1) * height);
  for (uint32_t y = 0; y < height; y++) {
    raw.push_back(0);
    raw.insert(raw.end(), rgba + y * row, rgba + (y + 1) * row);
  }
  // zlib stream of stored blocks
  std::vector<uint8_t> z {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
  size_t pos = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back(uint8_t(n & 0xFF)); z.push_back(uint8_t(n >> 8));
    z.push_back(uint8_t(~n & 0xFF)); z.push_back(uint8_t((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + (std::ptrdiff_t)pos, raw.begin() + (std::ptrdiff_t)(pos + n));
    pos += n;
  } while (pos < raw.size());
  put_u32(z, (b << 16) | a);
  put_chunk(f, crc, "IDAT", z);
  put_chunk(f, crc, "IEND", {});
  return bool(f);
}

bool write_depth_png(ReadBackData &image, const std::string &path) {
  uint32_t *words = reinterpret_cast<uint32_t *>(image.bytes.get());
  for (size_t i = 0; i < size_t(image.width) * image.height; i++) words[i] &= 0xffffffu;
  return write_png_rgba8(path, image.width, image.height, image.bytes.get());
}

bool write_rgba_png(ReadBackData &image, const std::string &path) {
  for (size_t i = 0; i < size_t(image.width) * image.height; i++) image.bytes[4 * i + 3] = 255;
  return write_png_rgba8(path, image.width, image.height, image.bytes.get());
}
