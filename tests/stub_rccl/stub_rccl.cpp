// stub_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/rccl_exchange.hip loads with dlopen
// (VKR_RCCL_LIBRARY selects this file), so that the native wire of the tiled frame — vkr_all_gather, vkr_all_gather_v,
// vkr_halo_exchange, the event ordering between the compute and the exchange stream, in-place gather offsets — runs with
// REAL PEER PROCESSES on a one-GPU box, where RCCL itself refuses two ranks on one device.
//
// Semantics kept: calls between ncclGroupStart / ncclGroupEnd form one operation that is enqueued on the caller's stream
// and completes in stream order; nothing blocks the host.  Transport: every rank's outgoing bytes are copied to pinned
// host memory on the stream, a host function (hipLaunchHostFunc, i.e. in stream order) moves them through per-pair
// mailboxes in a POSIX shared-memory segment named after the unique id, and the incoming bytes are copied back to the
// device on the stream.  Point-to-point messages of a pair arrive in the order they were sent and collectives are sets of
// such messages in rank order — so, as with real RCCL, the ranks must issue matching sends / receives / collectives in
// the same order per communicator, but a rank with nothing to exchange in a group need not take part in it.
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

// One mailbox per directed pair (s -> d): a single slot of `slot_bytes` and two sequence numbers.  Point-to-point
// messages of a pair are delivered in order (RCCL's rule); a message larger than the slot travels in chunks.  The ranks'
// groups need not line up: a rank whose group is empty simply does not touch any mailbox.
struct Mailbox {
  std::atomic<uint64_t> sent;      // chunks written by the sender
  std::atomic<uint64_t> consumed;  // chunks read by the receiver
  uint64_t chunk_bytes;            // size of the chunk in the slot
  uint64_t message_bytes;          // size of the message the chunk belongs to: must be what the receiver expects
  uint64_t pad[4];
};
struct Header {
  std::atomic<uint32_t> attached, detached;
  uint32_t world, pad;
  uint64_t slot_bytes;
  Mailbox box[MAX_RANKS][MAX_RANKS];
};

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double timeout_s() { const char* e = getenv("VKR_STUB_RCCL_TIMEOUT_S"); return e ? atof(e) : 120.0; }

struct Op { int kind; const void* send; void* recv; size_t bytes; int peer; };  // kind 0 allgather, 1 broadcast (peer = root), 2 send, 3 recv

struct Comm {
  int rank = 0, world = 1;
  std::string name;
  Header* hdr = nullptr;
  uint8_t* slots = nullptr;  // world x world slots of hdr->slot_bytes each, after the header
  size_t map_bytes = 0;
  uint64_t groups = 0;
  uint64_t stats[4] = {0, 0, 0, 0};  // bytes moved per kind
  uint64_t calls[4] = {0, 0, 0, 0};
  uint8_t* slot(int s, int d) const { return slots + ((size_t)s * world + d) * hdr->slot_bytes; }
};

struct Message { int peer; size_t staging_offset, bytes, done; };  // one point-to-point transfer; `done` bytes so far
struct Copy { void* dst; const void* src; size_t bytes; };
struct Group {
  Comm* comm;
  uint8_t* out_staging = nullptr; size_t out_bytes = 0;  // pinned: my outgoing bytes
  uint8_t* in_staging = nullptr; size_t in_bytes = 0;    // pinned: what I receive
  std::vector<Message> sends, recvs;                     // in issue order: per peer they are served first in, first out
  std::vector<Copy> self;                                // my own share of an all-gather / broadcast: staging to staging
  std::vector<Copy> landings;                            // in_staging -> device
  hipEvent_t done = nullptr;
};

thread_local bool t_in_group = false;
thread_local std::vector<std::pair<Comm*, Op>> t_ops;
thread_local hipStream_t t_stream = nullptr;
std::mutex g_retired_mutex;
std::vector<Group*> g_retired;

[[noreturn]] void die(const Comm* c, const Group* g) {
  fprintf(stderr, "[stub_rccl] rank %d/%d: no progress for %.0f s;", c->rank, c->world, timeout_s());
  for (const Message& m : g->sends) if (m.done < m.bytes) fprintf(stderr, " send->%d %zu/%zu", m.peer, m.done, m.bytes);
  for (const Message& m : g->recvs) if (m.done < m.bytes) fprintf(stderr, " recv<-%d %zu/%zu", m.peer, m.done, m.bytes);
  fprintf(stderr, "\n");
  fflush(stderr);
  _exit(3);
}

// runs in stream order on the exchange stream: my bytes are in out_staging, the device is waiting for in_staging.
// Progress engine: every directed pair is served first in, first out; sends and receives advance whenever their mailbox
// allows, so two ranks that both send before they receive (or send more than a slot holds) cannot block each other.
void host_exchange(void* arg) {
  Group* g = (Group*)arg;
  Comm* c = g->comm;
  Header* h = c->hdr;
  for (const Copy& s : g->self) std::memcpy(s.dst, s.src, s.bytes);
  size_t open = 0;
  for (const Message& m : g->sends) open += m.bytes ? 1 : 0;
  for (const Message& m : g->recvs) open += m.bytes ? 1 : 0;
  double last_progress = now_s();
  unsigned idle = 0;
  while (open) {
    bool moved = false;
    bool busy_to[MAX_RANKS] = {false}, busy_from[MAX_RANKS] = {false};  // only the oldest unfinished message of a pair may move
    for (Message& m : g->sends) {
      if (m.done == m.bytes || busy_to[m.peer]) continue;
      busy_to[m.peer] = true;
      Mailbox& b = h->box[c->rank][m.peer];
      if (b.sent.load(std::memory_order_relaxed) != b.consumed.load(std::memory_order_acquire)) continue;  // slot still full
      const size_t n = m.bytes - m.done < h->slot_bytes ? m.bytes - m.done : h->slot_bytes;
      std::memcpy(c->slot(c->rank, m.peer), g->out_staging + m.staging_offset + m.done, n);
      b.chunk_bytes = n;
      b.message_bytes = m.bytes;
      b.sent.store(b.sent.load(std::memory_order_relaxed) + 1, std::memory_order_release);
      m.done += n;
      if (m.done == m.bytes) --open;
      moved = true;
    }
    for (Message& m : g->recvs) {
      if (m.done == m.bytes || busy_from[m.peer]) continue;
      busy_from[m.peer] = true;
      Mailbox& b = h->box[m.peer][c->rank];
      if (b.sent.load(std::memory_order_acquire) == b.consumed.load(std::memory_order_relaxed)) continue;  // nothing there yet
      const size_t n = b.chunk_bytes;
      if (b.message_bytes != m.bytes || n > m.bytes - m.done) {
        fprintf(stderr, "[stub_rccl] rank %d: rank %d sent a chunk of %zu of a %llu-byte message where %zu of %zu were expected: the ranks' call sequences differ\n",
                c->rank, m.peer, n, (unsigned long long)b.message_bytes, m.bytes - m.done, m.bytes);
        _exit(5);
      }
      std::memcpy(g->in_staging + m.staging_offset + m.done, c->slot(m.peer, c->rank), n);
      b.consumed.store(b.consumed.load(std::memory_order_relaxed) + 1, std::memory_order_release);
      m.done += n;
      if (m.done == m.bytes) --open;
      moved = true;
    }
    if (moved) { last_progress = now_s(); idle = 0; continue; }
    if (++idle > 64) { usleep(50); if (now_s() - last_progress > timeout_s()) die(c, g); }
  }
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
  c->groups++;
  // Every collective is decomposed into point-to-point messages in the order every rank decomposes it, so the pairwise
  // first-in-first-out rule matches them up.  In-place operation (send inside recv) is fine: all sends are staged first.
  struct Out { const void* src; size_t bytes; size_t off; };
  struct In { void* dst; size_t bytes; size_t off; };
  std::vector<Out> outs;
  std::vector<In> ins;
  auto stage_out = [&](const void* src, size_t bytes) { outs.push_back({src, bytes, g->out_bytes}); g->out_bytes += bytes; return outs.back().off; };
  auto stage_in = [&](void* dst, size_t bytes) { ins.push_back({dst, bytes, g->in_bytes}); g->in_bytes += bytes; return ins.back().off; };
  struct SelfCopy { size_t in_off, out_off, bytes; };
  std::vector<SelfCopy> selfs;
  for (auto& po : t_ops) {
    const Op& o = po.second;
    c->calls[o.kind]++; c->stats[o.kind] += o.bytes;
    if (o.kind == 0) {  // all-gather: my share to everybody, everybody's share to me at [rank][bytes]
      const size_t so = stage_out(o.send, o.bytes);
      for (int r = 0; r < c->world; r++) {
        const size_t io = stage_in((uint8_t*)o.recv + (size_t)r * o.bytes, o.bytes);
        if (r == c->rank) { selfs.push_back({io, so, o.bytes}); continue; }
        g->sends.push_back({r, so, o.bytes, 0});
        g->recvs.push_back({r, io, o.bytes, 0});
      }
    } else if (o.kind == 1) {  // broadcast from root o.peer
      const size_t io = stage_in(o.recv, o.bytes);
      if (c->rank == o.peer) {
        const size_t so = stage_out(o.send, o.bytes);
        selfs.push_back({io, so, o.bytes});
        for (int r = 0; r < c->world; r++) if (r != c->rank) g->sends.push_back({r, so, o.bytes, 0});
      } else {
        g->recvs.push_back({o.peer, io, o.bytes, 0});
      }
    } else if (o.kind == 2) {
      g->sends.push_back({o.peer, stage_out(o.send, o.bytes), o.bytes, 0});
    } else {
      g->recvs.push_back({o.peer, stage_in(o.recv, o.bytes), o.bytes, 0});
    }
  }
  t_ops.clear();
  if (g->out_bytes && hipHostMalloc((void**)&g->out_staging, g->out_bytes, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
  if (g->in_bytes && hipHostMalloc((void**)&g->in_staging, g->in_bytes, hipHostMallocDefault) != hipSuccess) return ncclSystemError;
  for (const SelfCopy& s : selfs) g->self.push_back({g->in_staging + s.in_off, g->out_staging + s.out_off, s.bytes});
  for (const Out& o : outs)
    if (o.bytes && hipMemcpyAsync(g->out_staging + o.off, o.src, o.bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, host_exchange, g) != hipSuccess) return ncclUnhandledCudaError;
  for (const In& l : ins)
    if (l.bytes && hipMemcpyAsync(l.dst, g->in_staging + l.off, l.bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
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
  const char* mb = getenv("VKR_STUB_RCCL_SLOT_MB");
  const size_t slot_bytes = (size_t)(mb ? atoi(mb) : 8) << 20;
  c->map_bytes = sizeof(Header) + slot_bytes * nranks * nranks;  // sparse: only the pages a run touches exist
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
  c->slots = (uint8_t*)p + sizeof(Header);
  if (rank == 0) { c->hdr->world = (uint32_t)nranks; c->hdr->slot_bytes = slot_bytes; }
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
              (unsigned long long)c->groups, (unsigned long long)c->calls[0], (unsigned long long)c->stats[0], (unsigned long long)c->calls[1],
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
