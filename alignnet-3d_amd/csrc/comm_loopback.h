// In-process loopback communicator: W data-parallel "ranks" = W engine handles of ONE process on ONE device, each driven by its own host
// thread on its own stream.  It stands where RCCL stands (alignnet_train.hip: comm_allreduce / comm_allgather) so that the engine's
// multi-rank code -- sync_bn's per-layer sums, global_loss's gathers, the bucketed gradient all-reduce -- runs with DIFFERENT shards on a
// 1-GPU box.  (The reference is single-device, train.py:189; its batch statistics are over the whole batch, utils/tf_util.py:474, and its
// loss couples all samples, models/tp8.py:279,288,327 -- that is what a sharded step has to reproduce.)
//
// A collective is a host rendezvous of the W calling threads (the k-th collective call of every rank must be the same operation, as with
// RCCL) plus stream-ordered device work:
//   1. every rank records "my input is ready" on its stream and publishes its pointers;                      -- host barrier --
//   2. every rank's stream waits for all ready events and runs its share: all-reduce = the rank owns one slice of the vector, sums that
//      slice over all ranks' buffers in rank order (fixed order: every rank ends with bit-identical sums) and writes it back into all of
//      them (reduce-scatter + all-gather in one kernel, in place, no staging copy); all-gather = the rank pulls every rank's block into
//      its own output; then records "my share is done";                                                       -- host barrier --
//   3. every rank's stream waits for all done events (its buffer is complete / its input has been read by everyone).
// Events are per rank and reused: a rank re-records `ready` only after the second barrier of the previous collective (all waits on it have
// been issued) and `done` only after the first barrier of the next one.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

namespace alignnet {

constexpr int kLoopMaxWorld = 16;
// How long a rank waits at a rendezvous before it declares the group broken.  Generous on purpose: a peer may legitimately stall for a
// long time (first-step workspace allocation of 288 GB-sized shapes, a debugger, a profiler draining its buffers) and a timeout poisons
// the group for good; alignnet_set_option("loopback_timeout_s") on any rank of the group changes it.
inline std::chrono::seconds& loop_timeout() { static std::chrono::seconds t{600}; return t; }
constexpr char kLoopMagic[8] = {'A', 'L', 'N', '3', 'L', 'O', 'O', 'P'};   // first bytes of a loopback id (alignnet_comm_loopback_id)

struct LoopPtrs { void* p[kLoopMaxWorld]; };

template <typename T>
__global__ void loop_allreduce_kernel(const LoopPtrs tab, size_t lo, size_t hi, int world)
{
  const size_t i = lo + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= hi) return;
  T acc = static_cast<const T*>(tab.p[0])[i];
  for (int q = 1; q < world; ++q) acc += static_cast<const T*>(tab.p[q])[i];
  for (int q = 0; q < world; ++q) static_cast<T*>(tab.p[q])[i] = acc;
}

// dst[q * n4 + i] = src_q[i]  (four-byte elements)
__global__ void loop_allgather_kernel(const LoopPtrs src, unsigned* __restrict__ dst, size_t n4, int world)
{
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n4 * world) return;
  const int q = (int)(i / n4);
  dst[i] = static_cast<const unsigned*>(src.p[q])[i - (size_t)q * n4];
}

struct LoopGroup {
  int world = 0, device = -1;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  bool broken = false;
  int joined = 0;
  // what each rank published for the collective in flight
  void* ptr[kLoopMaxWorld] = {nullptr};
  size_t count[kLoopMaxWorld] = {0};
  int op[kLoopMaxWorld] = {0};
  hipEvent_t ready[kLoopMaxWorld] = {nullptr}, done[kLoopMaxWorld] = {nullptr};
  long long collectives = 0;   // completed rendezvous (rank 0's count)

  // all W threads meet here; false when the group is broken (a rank failed or did not show up within the timeout)
  bool barrier(std::unique_lock<std::mutex>& lk)
  {
    if (broken) return false;
    const unsigned long long my = gen;
    if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
    if (!cv.wait_for(lk, loop_timeout(), [&] { return gen != my || broken; })) { broken = true; cv.notify_all(); }
    return !broken;
  }
  void poison() { std::lock_guard<std::mutex> g(mu); broken = true; cv.notify_all(); }
};

struct LoopComm { std::shared_ptr<LoopGroup> g; int rank = 0; unsigned long long key = 0; };

inline std::mutex& loop_registry_mutex() { static std::mutex m; return m; }
inline std::map<unsigned long long, std::weak_ptr<LoopGroup>>& loop_registry() { static std::map<unsigned long long, std::weak_ptr<LoopGroup>> r; return r; }

inline bool loop_is_id(const unsigned char id[128]) { return std::memcmp(id, kLoopMagic, 8) == 0; }

inline void loop_make_id(unsigned char id[128])
{
  static std::mutex m; static unsigned long long next = 1;
  std::memset(id, 0, 128);
  std::memcpy(id, kLoopMagic, 8);
  std::lock_guard<std::mutex> g(m);
  const unsigned long long v = next++;
  std::memcpy(id + 8, &v, 8);
}

// joins (creating it on first use) the group named by `id`; returns nullptr + err on a mismatch
inline LoopComm* loop_join(const unsigned char id[128], int rank, int world, int device, std::string& err)
{
  if (world > kLoopMaxWorld) { err = "loopback communicator: at most " + std::to_string(kLoopMaxWorld) + " ranks"; return nullptr; }
  unsigned long long key; std::memcpy(&key, id + 8, 8);
  std::shared_ptr<LoopGroup> g;
  {
    std::lock_guard<std::mutex> lk(loop_registry_mutex());
    auto& reg = loop_registry();
    auto it = reg.find(key);
    if (it != reg.end()) g = it->second.lock();
    if (!g) { g = std::make_shared<LoopGroup>(); g->world = world; g->device = device; reg[key] = g; }
  }
  std::unique_lock<std::mutex> lk(g->mu);
  // every failure below breaks the group and wakes the peers: a rank that cannot join must not leave the others in the join
  // rendezvous until their timeout
  auto refuse = [&](const std::string& m) -> LoopComm* { err = m; g->broken = true; g->cv.notify_all(); return nullptr; };
  // (a group that is already complete is not the stray caller's to break: it is turned away, the joined ranks go on)
  auto turn_away = [&](const std::string& m) -> LoopComm* { if (g->joined == g->world && !g->broken) { err = m; return nullptr; } return refuse(m); };
  if (rank < 0 || rank >= world) return turn_away("loopback communicator: rank " + std::to_string(rank) + " outside the world of " + std::to_string(world));
  if (g->world != world) return turn_away("loopback communicator: ranks disagree on the world size");
  if (g->device != device) return turn_away("loopback communicator: all ranks must sit on one device");
  if (g->ready[rank]) return turn_away("loopback communicator: rank " + std::to_string(rank) + " joined twice");
  if (hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming) != hipSuccess) {
    if (g->ready[rank]) { hipEventDestroy(g->ready[rank]); g->ready[rank] = nullptr; }
    return refuse("loopback communicator: hipEventCreate failed");
  }
  g->joined++;
  g->cv.notify_all();
  // like ncclCommInitRank, joining is itself a rendezvous: nobody leaves before all `world` ranks are in
  if (!g->cv.wait_for(lk, loop_timeout(), [&] { return g->joined == g->world || g->broken; }) || g->broken) {
    const int have = g->joined;
    g->joined--;   // undo this rank's join: its events go, the (broken) group keeps no trace of it
    hipEventDestroy(g->ready[rank]); g->ready[rank] = nullptr;
    hipEventDestroy(g->done[rank]); g->done[rank] = nullptr;
    return refuse("loopback communicator: only " + std::to_string(have) + " of " + std::to_string(world) + " ranks joined");
  }
  LoopComm* c = new LoopComm();
  c->g = g; c->rank = rank; c->key = key;
  return c;
}

inline void loop_leave(LoopComm* c)
{
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(c->g->mu);
    c->g->broken = true;   // a group that lost a rank cannot complete another collective
    c->g->cv.notify_all();
    if (c->g->ready[c->rank]) { hipEventDestroy(c->g->ready[c->rank]); c->g->ready[c->rank] = nullptr; }
    if (c->g->done[c->rank]) { hipEventDestroy(c->g->done[c->rank]); c->g->done[c->rank] = nullptr; }
  }
  const unsigned long long key = c->key;
  delete c;   // (drops this rank's reference to the group)
  std::lock_guard<std::mutex> lk(loop_registry_mutex());   // the last rank to leave takes the expired entry out of the registry
  auto it = loop_registry().find(key);
  if (it != loop_registry().end() && it->second.expired()) loop_registry().erase(it);
}

enum { kLoopSumF32 = 1, kLoopSumF64 = 2, kLoopGather = 3 };

// op = kLoopSum*: in place on `buf` (n elements);  op = kLoopGather: `buf` = this rank's block (n four-byte elements), out = [world][n]
inline int loop_collective(LoopComm* c, int op, void* buf, void* out, size_t n, hipStream_t s, std::string& err)
{
  LoopGroup& g = *c->g;
  const int r = c->rank, W = g.world;
  LoopPtrs tab;
  {
    std::unique_lock<std::mutex> lk(g.mu);
    if (g.joined != W) { err = "loopback communicator: only " + std::to_string(g.joined) + " of " + std::to_string(W) + " ranks have joined"; g.broken = true; g.cv.notify_all(); return 1; }
    g.ptr[r] = buf; g.count[r] = n; g.op[r] = op;
    if (hipEventRecord(g.ready[r], s) != hipSuccess) { g.broken = true; g.cv.notify_all(); err = "loopback communicator: hipEventRecord failed"; return 1; }
    if (!g.barrier(lk)) { err = "loopback communicator: a rank failed or did not reach the collective (rendezvous broken)"; return 1; }
    for (int q = 0; q < W; ++q) {
      if (g.op[q] != op || g.count[q] != n) {
        g.broken = true; g.cv.notify_all();
        err = "loopback communicator: ranks issued different collectives (rank " + std::to_string(q) + ": op " + std::to_string(g.op[q]) + ", " +
              std::to_string(g.count[q]) + " elements; rank " + std::to_string(r) + ": op " + std::to_string(op) + ", " + std::to_string(n) + ")";
        return 1;
      }
      tab.p[q] = g.ptr[q];
    }
  }
  hipError_t e = hipSuccess;
  for (int q = 0; q < W && e == hipSuccess; ++q) if (q != r) e = hipStreamWaitEvent(s, g.ready[q], 0);
  if (e == hipSuccess && n) {
    if (op == kLoopGather) {
      hipLaunchKernelGGL(loop_allgather_kernel, dim3((unsigned)((n * W + 255) / 256)), dim3(256), 0, s, tab, static_cast<unsigned*>(out), n, W);
    } else {
      const size_t chunk = (n + W - 1) / W, lo = std::min(n, chunk * r), hi = std::min(n, lo + chunk);
      if (hi > lo) {
        const dim3 grid((unsigned)((hi - lo + 255) / 256));
        if (op == kLoopSumF64) hipLaunchKernelGGL(loop_allreduce_kernel<double>, grid, dim3(256), 0, s, tab, lo, hi, W);
        else hipLaunchKernelGGL(loop_allreduce_kernel<float>, grid, dim3(256), 0, s, tab, lo, hi, W);
      }
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(g.done[r], s);
  {
    std::unique_lock<std::mutex> lk(g.mu);
    if (e != hipSuccess) { g.broken = true; g.cv.notify_all(); err = std::string("loopback communicator: ") + hipGetErrorString(e); return 1; }
    if (!g.barrier(lk)) { err = "loopback communicator: a rank failed inside the collective (rendezvous broken)"; return 1; }
    if (r == 0) g.collectives++;
  }
  for (int q = 0; q < W && e == hipSuccess; ++q) if (q != r) e = hipStreamWaitEvent(s, g.done[q], 0);
  if (e != hipSuccess) { g.poison(); err = std::string("loopback communicator: ") + hipGetErrorString(e); return 1; }
  return 0;
}

}  // namespace alignnet
