// Peer-to-peer SyncBN statistics exchange over xGMI.
//
// The reference synchronises every BatchNorm of the step across the data-parallel ranks (nn.SyncBatchNorm in MlpDWBN,
// modules/ffn_block.py:222-234; train.sync_bn for the other 306, configs/base/loveda.py:106-108): ~660 exchanges of at most a few
// hundred floats per step, every one of them on the critical path.  Through a collective library each is a launch of a general
// all-reduce (10-20 us); here it is ONE single-workgroup kernel per rank:
//     1. fold the layer's slotted {sum, sumsq} (or {sum dz, sum dz*raw}) buffer to [2C] values,
//     2. store every value TOGETHER WITH THE EPOCH, as one 8-byte word, into the window of EVERY rank (xGMI peer writes into
//        hipIpc-mapped, fine-grained device memory),
//     3. spin on the own window until the words of all ranks carry this epoch (no separate flag, no fence: the NCCL "LL" idea),
//     4. add the contributions in RANK ORDER (every rank forms the same sum: replicas stay bit-identical) and write the total back
//        into slot 0 of the statistics buffer (the other slots are cleared: the consumers fold all slots).
// Windows are double-buffered by the parity of a per-channel epoch that lives in device memory and is advanced by the kernel
// itself, so the launch is replay-safe inside a captured hipGraph; a CHANNEL is an independent sequence of exchanges (one per stream
// that issues them: kernels of one channel run in stream order, kernels of different channels never touch the same flags).
// A bounded spin (rssf_p2p_set_timeout_ms; the start-up self-test and the tests use one) turns a missing peer into an error word
// instead of a hung GPU; in steady state the wait is unbounded, as a collective's is (a rank may legitimately be minutes late: it
// evaluates or writes a checkpoint while the others have entered the next step).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <new>
#include "common.hip.h"
using namespace rssf;

namespace {

constexpr int MAXF = RSSF_P2P_MAX_FLOATS;        // floats of one exchange (sum over its items of 2C)
constexpr int MAX_WORLD = 16;
typedef unsigned long long u64;

// window of one rank, per channel: slot[2 parities][world][MAXF] of 8 bytes = {value bits, epoch}
__host__ __device__ inline size_t chan_slots(int world) { return (size_t)2 * world * MAXF; }

struct ExArgs {
  u64* win[MAX_WORLD];          // base of every rank's window (this process's mapping), already offset to the channel
  unsigned* epoch;              // this rank's epoch counter of the channel (device memory)
  unsigned* err;                // this rank's error word: != 0 after a timed-out wait
  u64* wait;                    // this rank's {ticks spent waiting for peers, exchanges} of the channel (rssf_p2p_wait_us)
  float* stats;
  int item_off[RSSF_P2P_MAX_ITEMS], item_n[RSSF_P2P_MAX_ITEMS];   // per layer: offset of its [nslots][n] block in `stats`, n = 2C
  int nitems, nslots, rank, world;
  long long timeout_ticks;      // wall_clock64 ticks (100 MHz)
};

// "LL" exchange: value and epoch travel in ONE 8-byte store, so a reader that sees the epoch it waits for has the value - no flag,
// no fence, no cache maintenance (a system-scope release / __threadfence_system() writes the L2 back: microseconds on this part,
// DESIGN.md lesson 2).  Relaxed system-scope atomics on fine-grained memory go to memory / the fabric directly.
__global__ void __launch_bounds__(256) p2p_exchange_kernel(ExArgs a) {
  const int tid = threadIdx.x;
  const unsigned e = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;     // uniform: advanced at the end
  const int par = (int)(e & 1u);
  int F = 0;
  for (int i = 0; i < a.nitems; ++i) F += a.item_n[i];
  const size_t mine = ((size_t)par * a.world + a.rank) * MAXF;
  const u64* inbox = a.win[a.rank] + (size_t)par * a.world * MAXF;
  for (int f = tid; f < F; f += 256) {
    int i = 0, j = f;
    while (j >= a.item_n[i]) { j -= a.item_n[i]; ++i; }
    float* blk = a.stats + a.item_off[i] + j;
    const int n = a.item_n[i];
    // the slots in slot order; all loads of a pass in flight together (as a loop the sixteen loads were sixteen round trips in a row)
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= a.nslots; s += 8) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = blk[(size_t)(s + k) * n];
#pragma unroll
      for (int k = 0; k < 8; ++k) v += x[k];
    }
    for (; s < a.nslots; ++s) v += blk[(size_t)s * n];
    const u64 word = ((u64)e << 32) | (u64)__float_as_uint(v);
    // (this rank's own contribution stays in the register: no round trip through its uncached window)
    for (int r = 0; r < a.world; ++r)
      if (r != a.rank) __hip_atomic_store(a.win[r] + mine + f, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // rank-ordered sum of what the ranks sent (every rank adds in the same order: bit-identical totals)
    float t = 0.f;
    const long long t0 = wall_clock64();
    for (int r = 0; r < a.world; ++r) {
      if (r == a.rank) { t += v; continue; }
      u64 w = __hip_atomic_load(inbox + (size_t)r * MAXF + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      while ((unsigned)(w >> 32) != e) {
        if (a.timeout_ticks > 0 && wall_clock64() - t0 > a.timeout_ticks) { atomicExch(a.err, 1u + (unsigned)r); break; }
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(inbox + (size_t)r * MAXF + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      t += __uint_as_float((unsigned)w);
    }
    blk[0] = t;
    for (int s = 1; s < a.nslots; ++s) blk[(size_t)s * n] = 0.f;
    // what this exchange WAITED for its peers (value 0 of the exchange: the words of one sender arrive together), for the per-rank
    // diagnosis of a multi-GPU run (bench.py --gpus N): kernels of one channel run in stream order, so a plain read-modify-write is enough
    if (f == 0) { a.wait[0] += (u64)(wall_clock64() - t0); a.wait[1] += 1; }
  }
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

struct rssf_p2p {
  int rank, world, channels;
  u64* window;                   // own window: channels * chan_slots(world) 8-byte slots, fine-grained
  u64* peer[MAX_WORLD];        // mapped windows (peer[rank] == window)
  bool opened[MAX_WORLD];
  unsigned* counters;            // [channels] epochs + [1] error word (ordinary device memory: only this rank's kernels touch it)
  u64* waits;                    // [channels][2] {wall_clock64 ticks waited for peers, exchanges}
  long long timeout_ticks;
};

extern "C" int rssf_p2p_create(rssf_p2p** out, int rank, int world, int channels, void* ipc_handle64) {
  RSSF_REQUIRE(out && ipc_handle64 && world >= 1 && world <= MAX_WORLD && rank >= 0 && rank < world && channels >= 1 && channels <= 16,
               "p2p_create: bad arguments (rank %d, world %d, channels %d)", rank, world, channels);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries the hipIpc handle as 64 opaque bytes");
  rssf_p2p* h = new (std::nothrow) rssf_p2p();
  RSSF_REQUIRE(h, "p2p_create: out of memory");
  memset(h, 0, sizeof(*h));
  h->rank = rank; h->world = world; h->channels = channels;
  const size_t bytes = (size_t)channels * chan_slots(world) * sizeof(u64);
  // fine-grained (uncached) device memory: peers' stores and this rank's loads are coherent INSIDE a running kernel
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&h->window), bytes, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(h->window, 0, bytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->counters), (channels + 1) * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(h->counters, 0, (channels + 1) * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->waits), (size_t)channels * 2 * sizeof(u64));
  if (e == hipSuccess) e = hipMemset(h->waits, 0, (size_t)channels * 2 * sizeof(u64));
  hipIpcMemHandle_t ih;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&ih, h->window);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error("p2p_create: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    if (h->window) (void)hipFree(h->window);
    if (h->counters) (void)hipFree(h->counters);
    if (h->waits) (void)hipFree(h->waits);
    delete h;
    return RSSF_ERR_LAUNCH;
  }
  memcpy(ipc_handle64, &ih, 64);
  h->peer[rank] = h->window;
  h->timeout_ticks = 10000LL * 100000LL;       // 10 s at wall_clock64's 100 MHz until rssf_p2p_set_timeout_ms says otherwise
  *out = h;
  return RSSF_OK;
}

extern "C" int rssf_p2p_connect(rssf_p2p* h, int peer, const void* ipc_handle64) {
  RSSF_REQUIRE(h && ipc_handle64 && peer >= 0 && peer < h->world, "p2p_connect: bad arguments");
  if (peer == h->rank || h->peer[peer]) return RSSF_OK;
  hipIpcMemHandle_t ih;
  memcpy(&ih, ipc_handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    set_error("p2p_connect: hipIpcOpenMemHandle(rank %d): %s", peer, hipGetErrorString(e));
    (void)hipGetLastError();
    return RSSF_ERR_LAUNCH;
  }
  h->peer[peer] = reinterpret_cast<u64*>(p);
  h->opened[peer] = true;
  return RSSF_OK;
}

// Ranks that live in ONE process (one rank per stream or per device of the process: hipIpc handles cannot be opened by the process
// that exported them): rank `peer` is the object `other`, its window is addressed directly.
extern "C" int rssf_p2p_connect_local(rssf_p2p* h, int peer, rssf_p2p* other) {
  RSSF_REQUIRE(h && other && peer >= 0 && peer < h->world && other->rank == peer && other->world == h->world && other->channels == h->channels,
               "p2p_connect_local: bad arguments");
  if (peer == h->rank || h->peer[peer]) return RSSF_OK;
  h->peer[peer] = other->window;          // not `opened`: the owner frees it
  return RSSF_OK;
}

extern "C" int rssf_p2p_exchange(rssf_p2p* h, int channel, float* stats, const int* item_off, const int* item_n, int nitems, int nslots,
                                 void* stream) {
  RSSF_REQUIRE(h && stats && item_off && item_n && channel >= 0 && channel < h->channels && nitems >= 1 && nitems <= RSSF_P2P_MAX_ITEMS &&
                   nslots >= 1,
               "p2p_exchange: bad arguments (channel %d, items %d, slots %d)", channel, nitems, nslots);
  ExArgs a;
  int F = 0;
  for (int i = 0; i < nitems; ++i) {
    RSSF_REQUIRE(item_n[i] > 0 && item_off[i] >= 0, "p2p_exchange: bad item %d", i);
    a.item_off[i] = item_off[i]; a.item_n[i] = item_n[i];
    F += item_n[i];
  }
  RSSF_REQUIRE(F <= MAXF, "p2p_exchange: %d floats exceed the window (%d)", F, MAXF);
  for (int r = 0; r < h->world; ++r) {
    RSSF_REQUIRE(h->peer[r], "p2p_exchange: rank %d is not connected", r);
    a.win[r] = h->peer[r] + (size_t)channel * chan_slots(h->world);
  }
  a.epoch = h->counters + channel;
  a.err = h->counters + h->channels;
  a.wait = h->waits + (size_t)channel * 2;
  a.stats = stats; a.nitems = nitems; a.nslots = nslots; a.rank = h->rank; a.world = h->world;
  a.timeout_ticks = h->timeout_ticks;
  p2p_exchange_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a);
  return check_launch("p2p_exchange");
}

extern "C" int rssf_p2p_set_timeout_ms(rssf_p2p* h, int ms) {
  RSSF_REQUIRE(h && ms >= 0, "p2p_set_timeout_ms: bad arguments");
  h->timeout_ticks = (long long)ms * 100000LL;       // 0: wait for the peers for ever, like a collective library does
  return RSSF_OK;
}

extern "C" int rssf_p2p_status(rssf_p2p* h, int* timed_out) {
  RSSF_REQUIRE(h && timed_out, "p2p_status: bad arguments");
  unsigned v = 0;
  const hipError_t e = hipMemcpy(&v, h->counters + h->channels, sizeof(v), hipMemcpyDeviceToHost);      // synchronises
  if (e != hipSuccess) { set_error("p2p_status: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  *timed_out = (int)v;
  return RSSF_OK;
}

extern "C" int rssf_p2p_wait_us(rssf_p2p* h, int channel, double* wait_us, int64_t* exchanges, int reset) {
  RSSF_REQUIRE(h && channel >= 0 && channel < h->channels && wait_us && exchanges, "p2p_wait_us: bad arguments");
  u64 v[2] = {0, 0};
  hipError_t e = hipMemcpy(v, h->waits + (size_t)channel * 2, sizeof(v), hipMemcpyDeviceToHost);      // synchronises
  if (e == hipSuccess && reset) e = hipMemset(h->waits + (size_t)channel * 2, 0, sizeof(v));
  if (e != hipSuccess) { set_error("p2p_wait_us: %s", hipGetErrorString(e)); return RSSF_ERR_LAUNCH; }
  *wait_us = (double)v[0] * 0.01;          // wall_clock64: 100 MHz
  *exchanges = (int64_t)v[1];
  return RSSF_OK;
}

extern "C" int rssf_p2p_destroy(rssf_p2p* h) {
  if (!h) return RSSF_OK;
  for (int r = 0; r < h->world; ++r)
    if (h->opened[r] && h->peer[r]) (void)hipIpcCloseMemHandle(h->peer[r]);
  if (h->window) (void)hipFree(h->window);
  if (h->counters) (void)hipFree(h->counters);
  if (h->waits) (void)hipFree(h->waits);
  (void)hipGetLastError();
  delete h;
  return RSSF_OK;
}
