// stub_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/rccl_exchange.hip loads with dlopen
// (VKR_RCCL_LIBRARY selects this file), so that the native wire of the tiled frame — vkr_all_gather, vkr_all_gather_v,
// vkr_halo_exchange, the event ordering between the compute and the exchange stream, in-place gather offsets — runs with
// REAL PEER PROCESSES on a one-GPU box, where RCCL itself refuses two ranks on one device.
//
// Semantics kept: calls between ncclGroupStart / ncclGroupEnd form one operation that is enqueued on the caller's stream
// and completes in stream order; nothing blocks the host.  Transport: every rank's outgoing bytes are copied to pinned
// host memory on the stream, a host function (hipLaunchHostFunc, i.e. in stream order) publishes them in a POSIX
// shared-memory segment named after the unique id and collects what its peers published, and the incoming bytes are
// copied back to the device on the stream.  Group g of one rank pairs with group g of every other rank — the same
// requirement real RCCL has (all ranks issue the same sequence of collectives per communicator).
//
// A peer that does not show up within VKR_STUB_RCCL_TIMEOUT_S (default 120) makes the waiting rank print what it was
// waiting for and _exit(3): a hung exchange ends as a failed process, never as a hung test.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr int MAX_ENTRIES = 256;
constexpr int DST_ALL = -1;

struct Entry { int32_t dst; uint32_t pad; uint64_t offset, bytes; };
struct RankBox {
  std::atomic<uint64_t> posted;    // generation of the group whose bytes are in the outbox
  std::atomic<uint64_t> consumed;  // generation this rank has finished reading from everybody
  uint32_t entry_count, pad;
  Entry entries[MAX_ENTRIES];
};
struct Header {
  std::atomic<uint32_t> attached, detached;
  uint32_t world, pad;
  uint64_t outbox_bytes;
  RankBox box[MAX_RANKS];
};

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double timeout_s() { const char* e = getenv("VKR_STUB_RCCL_TIMEOUT_S"); return e ? atof(e) : 120.0; }

struct Op { int kind; const void* send; void* recv; size_t bytes; int peer; };  // kind 0 allgather, 1 broadcast (peer = root), 2 send, 3 recv

struct Comm {
  int rank = 0, world = 1;
  std::string name;
  Header* hdr = nullptr;
  uint8_t* outbox_base = nullptr;  // world outboxes of hdr->outbox_bytes each, after the header
  size_t map_bytes = 0;
  uint64_t generation = 0;
  uint64_t stats[4] = {0, 0, 0, 0};  // bytes moved per kind
  uint64_t calls[4] = {0, 0, 0, 0};
  uint8_t* outbox(int r) const { return outbox_base + (size_t)r * hdr->outbox_bytes; }
};

struct Landing { void* dst; size_t bytes; size_t staging_offset; int from; int match; };  // match: k-th entry of `from` addressed to me / to all
struct Group {
  Comm* comm;
  uint64_t generation;
  uint8_t* out_staging = nullptr; size_t out_bytes = 0;  // pinned: my outgoing bytes, in op order
  std::vector<Entry> out_entries;
  uint8_t* in_staging = nullptr; size_t in_bytes = 0;    // pinned: what I receive, in landing order
  std::vector<Landing> landings;
  hipEvent_t done = nullptr;
};

thread_local bool t_in_group = false;
thread_local std::vector<std::pair<Comm*, Op>> t_ops;
thread_local hipStream_t t_stream = nullptr;
std::mutex g_retired_mutex;
std::vector<Group*> g_retired;

[[noreturn]] void die(const Comm* c, const char* what, int peer, uint64_t gen) {
  fprintf(stderr, "[stub_rccl] rank %d/%d: timed out after %.0f s waiting for %s of rank %d (group %llu)\n", c->rank, c->world, timeout_s(), what,
          peer, (unsigned long long)gen);
  fflush(stderr);
  _exit(3);
}

void wait_at_least(const Comm* c, const std::atomic<uint64_t>& v, uint64_t want, const char* what, int peer) {
  const double t0 = now_s();
  unsigned spins = 0;
  while (v.load(std::memory_order_acquire) < want) {
    if (++spins > 64) { usleep(50); if (now_s() - t0 > timeout_s()) die(c, what, peer, want); }
  }
}

// runs in stream order on the exchange stream: my bytes are in out_staging, the device is waiting for in_staging
void host_exchange(void* arg) {
  Group* g = (Group*)arg;
  Comm* c = g->comm;
  Header* h = c->hdr;
  const uint64_t gen = g->generation;
  // 1. everybody has read my previous outbox
  for (int r = 0; r < c->world; r++) wait_at_least(c, h->box[r].consumed, gen - 1, "the previous group to be consumed", r);
  // 2. publish
  RankBox& mine = h->box[c->rank];
  if (g->out_bytes > h->outbox_bytes || g->out_entries.size() > MAX_ENTRIES) {
    fprintf(stderr, "[stub_rccl] rank %d: group of %zu bytes / %zu entries exceeds the outbox (VKR_STUB_RCCL_MB)\n", c->rank, g->out_bytes, g->out_entries.size());
    _exit(4);
  }
  if (g->out_bytes) std::memcpy(c->outbox(c->rank), g->out_staging, g->out_bytes);
  mine.entry_count = (uint32_t)g->out_entries.size();
  for (size_t i = 0; i < g->out_entries.size(); i++) mine.entries[i] = g->out_entries[i];
  mine.posted.store(gen, std::memory_order_release);
  // 3. collect
  for (const Landing& l : g->landings) {
    const RankBox& src = h->box[l.from];
    wait_at_least(c, src.posted, gen, "a group to be posted", l.from);
    int seen = 0;
    const Entry* hit = nullptr;
    for (uint32_t i = 0; i < src.entry_count; i++) {
      const Entry& e = src.entries[i];
      if (e.dst == DST_ALL || e.dst == c->rank) { if (seen == l.match) { hit = &e; break; } ++seen; }
    }
    if (!hit || hit->bytes != l.bytes) {
      fprintf(stderr, "[stub_rccl] rank %d: group %llu: rank %d published %s for landing %d (want %zu bytes, got %llu): the ranks' call sequences differ\n",
              c->rank, (unsigned long long)gen, l.from, hit ? "a different size" : "nothing", l.match, l.bytes, hit ? (unsigned long long)hit->bytes : 0ull);
      _exit(5);
    }
    std::memcpy(g->in_staging + l.staging_offset, c->outbox(l.from) + hit->offset, l.bytes);
  }
  // 4. done with everybody's outbox of this generation
  mine.consumed.store(gen, std::memory_order_release);
}

void reap_retired() {
  std::lock_guard<std::mutex> lock(g_retired_mutex);
  for (size_t i = 0; i < g_retired.size();) {
    Group* g = g_retired[i];
    if (hipEventQuery(g->done) == hipSuccess) {
      (void)hipEventDestroy(g->done);
      if (g->out_staging) (void)hipHostFree(g->out_staging);
      if (g->in_staging) (void)hipHostFree(g->in_staging);
      delete g;
      g_retired[i] = g_retired.back();
      g_retired.pop_back();
    } else {
      ++i;
    }
  }
}

ncclResult_t run_group(hipStream_t stream) {
  if (t_ops.empty()) return ncclSuccess;
  Comm* c = t_ops[0].first;
  for (auto& o : t_ops) if (o.first != c) return ncclInvalidUsage;  // one communicator per group is all the tiled frame needs
  reap_retired();
  Group* g = new Group;
  g->comm = c;
  g->generation = ++c->generation;
  // layout of my outbox and of my landings; in-place operation (send inside recv) is fine: all sends are staged first
  struct Out { const void* src; size_t bytes; size_t off; };
  std::vector<Out> outs;
  std::vector<int> seen_from(c->world, 0);  // entries of rank r addressed to me (or all) matched so far
  for (auto& po : t_ops) {
    const Op& o = po.second;
    c->calls[o.kind]++; c->stats[o.kind] += o.bytes;
    if (o.kind == 0) {  // all-gather: my share to everybody, everybody's share to me at [rank][bytes]
      outs.push_back({o.send, o.bytes, g->out_bytes});
      g->out_entries.push_back({DST_ALL, 0, g->out_bytes, o.bytes});
      g->out_bytes += o.bytes;
      for (int r = 0; r < c->world; r++) {
        g->landings.push_back({(uint8_t*)o.recv + (size_t)r * o.bytes, o.bytes, g->in_bytes, r, seen_from[r]++});
        g->in_bytes += o.bytes;
      }
    } else if (o.kind == 1) {  // broadcast from root o.peer
      if (c->rank == o.peer) {
        outs.push_back({o.send, o.bytes, g->out_bytes});
        g->out_entries.push_back({DST_ALL, 0, g->out_bytes, o.bytes});
        g->out_bytes += o.bytes;
      }
      g->landings.push_back({o.recv, o.bytes, g->in_bytes, o.peer, seen_from[o.peer]++});
      g->in_bytes += o.bytes;
    } else if (o.kind == 2) {
      outs.push_back({o.send, o.bytes, g->out_bytes});
      g->out_entries.push_back({o.peer, 0, g->out_bytes, o.bytes});
      g->out_bytes += o.bytes;
    } else {
      g->landings.push_back({o.recv, o.bytes, g->in_bytes, o.peer, seen_from[o.peer]++});
      g->in_bytes += o.bytes;
    }
  }
  // NOTE on `match`: an all-gather / broadcast entry counts for every reader, a send only for its destination, and each
  // reader counts the entries it can see in the publisher's order — both sides walk the same list, so the k-th landing
  // from rank r is the k-th visible entry of rank r.
  t_ops.clear();
  if (g->out_bytes && hipHostMalloc((void**)&g->out_staging, g->out_bytes, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
  if (g->in_bytes && hipHostMalloc((void**)&g->in_staging, g->in_bytes, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
  for (const Out& o : outs)
    if (hipMemcpyAsync(g->out_staging + o.off, o.src, o.bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, host_exchange, g) != hipSuccess) return ncclUnhandledCudaError;
  for (const Landing& l : g->landings)
    if (hipMemcpyAsync(l.dst, g->in_staging + l.staging_offset, l.bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventCreateWithFlags(&g->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(g->done, stream) != hipSuccess) return ncclUnhandledCudaError;
  std::lock_guard<std::mutex> lock(g_retired_mutex);
  g_retired.push_back(g);
  return ncclSuccess;
}

ncclResult_t add_op(ncclComm_t comm, Op op, hipStream_t stream) {
  t_ops.emplace_back((Comm*)comm, op);
  t_stream = stream;
  if (!t_in_group) return run_group(stream);
  return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/vkrstub_%d_%llx", (int)getpid(), (unsigned long long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank; c->world = nranks;
  c->name = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
  const char* mb = getenv("VKR_STUB_RCCL_MB");
  const size_t outbox_bytes = (size_t)(mb ? atoi(mb) : 64) << 20;
  c->map_bytes = sizeof(Header) + outbox_bytes * nranks;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
  } else {
    const double t0 = now_s();
    struct stat st;
    for (;;) {  // rank 0 creates and sizes the segment
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
      if (fd >= 0) close(fd);
      if (now_s() - t0 > timeout_s()) { delete c; return ncclSystemError; }
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->hdr = (Header*)p;  // a fresh segment is zero-filled: every counter starts at 0
  c->outbox_base = (uint8_t*)p + sizeof(Header);
  if (rank == 0) { c->hdr->world = (uint32_t)nranks; c->hdr->outbox_bytes = outbox_bytes; }
  c->hdr->attached.fetch_add(1);
  const double t0 = now_s();
  while (c->hdr->attached.load() < (uint32_t)nranks) {  // collective, like the real call
    if (now_s() - t0 > timeout_s()) { fprintf(stderr, "[stub_rccl] rank %d: peers never attached\n", rank); _exit(3); }
    usleep(200);
  }
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  (void)hipDeviceSynchronize();
  reap_retired();
  if (const char* log = getenv("VKR_STUB_RCCL_LOG")) {  // what actually crossed the wire, for the tests to assert on
    if (FILE* f = fopen(log, "a")) {
      fprintf(f, "rank %d world %d groups %llu allgather %llu/%llu broadcast %llu/%llu send %llu/%llu recv %llu/%llu\n", c->rank, c->world,
              (unsigned long long)c->generation, (unsigned long long)c->calls[0], (unsigned long long)c->stats[0], (unsigned long long)c->calls[1],
              (unsigned long long)c->stats[1], (unsigned long long)c->calls[2], (unsigned long long)c->stats[2], (unsigned long long)c->calls[3],
              (unsigned long long)c->stats[3]);
      fclose(f);
    }
  }
  const bool last = c->hdr->detached.fetch_add(1) + 1 == (uint32_t)c->world;
  munmap(c->hdr, c->map_bytes);
  if (last) shm_unlink(c->name.c_str());
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { t_in_group = true; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { t_in_group = false; return run_group(t_stream); }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
  return add_op(comm, Op{0, send, recv, count * type_bytes(type), -1}, stream);
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
  return add_op(comm, Op{1, send, recv, count * type_bytes(type), root}, stream);
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return add_op(comm, Op{2, send, nullptr, count * type_bytes(type), peer}, stream);
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return add_op(comm, Op{3, nullptr, recv, count * type_bytes(type), peer}, stream);
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "stub_rccl error"; }

}  // extern "C"
