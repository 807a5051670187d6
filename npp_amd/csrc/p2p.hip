// One-shot SyncBatchNorm statistics exchange between the GPUs of ONE node (npp_p2p_*): every rank stores its vector straight
// into a mailbox in each peer's HBM (xGMI peer stores through hipIpc-mapped pointers), polls its own mailbox and sums the world's
// vectors in rank order -- one small kernel per exchange instead of an RCCL all-reduce.
//
// Why: the reference converts every BatchNorm to SyncBatchNorm (augment_lip_sync.py:191, search_lip_sync.py:268-271); a training
// step of model_augment then carries ~450-980 exchanges of a few KiB, each one a link of the dependent kernel chain.  A ring / tree
// all-reduce spends 2 (W - 1) (ring) or 2 log W hops of xGMI latency plus RCCL's launch and proxy overheads on each; here it
// is ONE hop: W - 1 peer writes in parallel and a poll.  The sum is taken in rank order on every rank, so all ranks hold
// bit-identical statistics (as after an all-reduce), and the kernel is an ordinary launch: capturable on ANY stream, no
// communicator-wide ordering between streams (each channel has mailboxes and a sequence counter of its own).
//
// Wire format (round 4; the round-3 form sent the data, waited for the stores and then raised a separate flag -- an ordering
// between two different addresses that nothing on one device could prove for xGMI): the "LL" protocol RCCL itself uses on these
// links.  A double travels as TWO 8-byte units {32 data bits | 32-bit sequence tag}; a unit is one naturally aligned 8-byte
// system-scope atomic store, which the fabric never splits, so a reader that sees the tag of exchange s in a unit holds that
// exchange's data bits of the unit -- no flag, no fence, no store -> flag ordering to rely on.
//
//   mailbox of rank r, channel c:   data [SLOTS][world][cap] x 2 units
//   exchange number s of a channel (device counter, so that hipGraph replays advance it) carries tag (u32)(s + 1) and uses slot
//   s % SLOTS.  A rank can only start exchange s + 1 after it has read every peer's units of s, i.e. after every peer has WRITTEN
//   s; a peer writes s + 2 only after it has read everybody's s + 1, which they send after finishing their reads of s (one kernel
//   after the other on the channel's stream): two slots would do, four are used.  A unit whose tag is AHEAD of the expected one
//   means a peer overwrote the slot before this rank read it (only possible after that peer gave up on a time-out): error bit 2.
//   Memory: uncached when the runtime offers it, else fine-grained, else plain device memory (the units then carry release /
//   acquire semantics).
//   A poll that sees nothing for NPP_P2P_TIMEOUT_MS (default 120 000, the order of an NCCL watchdog rather than of a step) gives
//   up: error bit 1.  After ANY error the channel is dead: its exchanges no longer wait and return NaN sums, so the statistics,
//   the loss and every gradient of the step are visibly void (never silently local), the GPU is not left spinning, and the host
//   (npp_p2p_status; train_step.TrainStep checks it every few steps with a MAX all-reduce of the error bits and raises) ends the run.
// RCCL (npp_syncbn_exchange) stays the transport for anything this does not cover: several nodes, vectors above the mailbox
// capacity, a runtime without peer access.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "p2p_xp.h"      // P2P_SLOTS, P2P_MAX_WORLD; the in-kernel form of the exchange (the fused BatchNorm kernels of bn.hip)

namespace {

constexpr int P2P_MAX_CHANNELS = 4;
constexpr int P2P_MAX_BLOCKS = 8;      // workgroups of one exchange: each owns an interleaved share of the vector

struct Channel {
  // (device pointers) mailbox of every rank as mapped into THIS process; [me] is the local allocation
  unsigned long long* data[P2P_MAX_WORLD];      // [SLOTS][world][cap][2 units]
  unsigned long long* seq;       // local: exchange counter
  unsigned int* err;             // local: bit 0 = a poll timed out, bit 1 = a peer overwrote a slot this rank had not read yet
};

struct P2P {
  int rank = -1, world = 0;
  long cap = 0;                  // doubles per (slot, source rank)
  int nchan = 0;
  void* local = nullptr;         // this rank's allocation: nchan mailboxes + the local words
  size_t bytes = 0;
  void* peers[P2P_MAX_WORLD] = {};
  Channel ch[P2P_MAX_CHANNELS];
  long long timeout_ticks = 0;
  int alloc_kind = -1;           // 0 uncached, 1 fine-grained, 2 plain device memory
  unsigned long long* res = nullptr;      // [nchan][cap][2 units]: the result vectors of the in-kernel exchanges (p2p_xp.h), plain device memory
} g;

size_t mailbox_bytes(long cap, int world) {
  size_t b = (size_t)P2P_SLOTS * world * cap * 2 * sizeof(unsigned long long);      // two {data32 | tag32} units per double
  return (b + 255) & ~(size_t)255;
}
size_t local_words_off(long cap, int world, int nchan) { return mailbox_bytes(cap, world) * nchan; }
size_t total_bytes(long cap, int world, int nchan) { return local_words_off(cap, world, nchan) + 256 * (size_t)nchan; }

void map_channels(void* base, int r, long cap, int world, int nchan, bool local) {
  for (int c = 0; c < nchan; ++c) {
    char* mb = static_cast<char*>(base) + mailbox_bytes(cap, world) * c;
    g.ch[c].data[r] = reinterpret_cast<unsigned long long*>(mb);
    if (local) {
      char* lw = static_cast<char*>(base) + local_words_off(cap, world, nchan) + 256 * (size_t)c;
      g.ch[c].seq = reinterpret_cast<unsigned long long*>(lw);
      g.ch[c].err = reinterpret_cast<unsigned int*>(lw + 64);
    }
  }
}

constexpr int P2P_MAX_SEGS = 8;
struct ExArgs {
  unsigned long long* peer_data[P2P_MAX_WORLD];
  unsigned long long* seq;       // [0] exchange counter, [1] workgroups of the current exchange that are done
  unsigned int* err;
  double* v;                     // plain form: the vector (nseg == 0)
  long n, cap;
  int me, world;
  long long timeout_ticks;
  int light;                     // uncached / fine-grained mailboxes: relaxed units (else release / acquire units)
  // slab form (npp_p2p_exchange_slabs): the vector is the concatenation of nseg segments, segment k = sum over its nrep replica
  // slabs [nrep][len]; the local sums also go to out_a (elements [0, split)) / out_b ([split, len)) as floats; the world's sum
  // lands in replica 0
  int nseg;
  double* seg[P2P_MAX_SEGS];
  long seg_len[P2P_MAX_SEGS], seg_split[P2P_MAX_SEGS];
  int seg_nrep[P2P_MAX_SEGS], seg_zero[P2P_MAX_SEGS];
  float* seg_out[P2P_MAX_SEGS][4];      // [0], [1], [2]: elements [0, split), [split, 2 split), [2 split, 3 split); [3]: a second copy of [0]
  float* seg_all[P2P_MAX_SEGS];         // every local sum of the segment (len floats), or NULL
};

NPP_DEV void seg_of(const ExArgs& a, long i, int& k, long& j) {
  k = 0; j = i;
  while (k + 1 < a.nseg && j >= a.seg_len[k]) { j -= a.seg_len[k]; ++k; }
}

// Workgroup b of B handles the elements i = b * 1024 + t, + B * 1024, ...: it pushes them to every mailbox, polls its own mailbox for
// the world's units of the same elements and sums them -- no synchronisation between the workgroups of a launch and none between
// push and poll.  The exchange counter moves when the last workgroup finishes (every workgroup has read it by then).
__global__ __launch_bounds__(1024) void p2p_exchange_kernel(ExArgs a) {
  const int t = threadIdx.x, b = blockIdx.x, B = gridDim.x;
  const unsigned long long s = __hip_atomic_load(a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned tag = (unsigned)(s + 1);
  const unsigned long long tagw = (unsigned long long)tag << 32;
  const int slot = (int)(s % P2P_SLOTS);
  const long off = ((long)slot * a.world + a.me) * a.cap;
  const long first = (long)b * 1024 + t, step = (long)B * 1024;
  // push: my share of the vector into slot [slot][me] of every mailbox (mine included), each double as two tagged 8-byte units
  for (long i = first; i < a.n; i += step) {
    double v;
    if (a.nseg == 0) {
      v = a.v[i];
    } else {
      int k; long j;
      seg_of(a, i, k, j);
      const double* sl = a.seg[k];
      const long len = a.seg_len[k];
      v = 0.0;
      const int nrep = a.seg_nrep[k];
      if (nrep <= 16) {      // (NPP_STAT_REPLICAS slabs: all loads in flight together)
        double part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = r < nrep ? sl[(long)r * len + j] : 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += part[r];
      } else {
        for (int r = 0; r < nrep; ++r) v += sl[(long)r * len + j];
      }
      const long sp = a.seg_split[k];
      const int part = sp > 0 ? (int)(j / sp) : 0;
      if (part < 3 && a.seg_out[k][part]) a.seg_out[k][part][j - part * sp] = (float)v;      // (elements past 3 * split go nowhere)
      if (part == 0 && a.seg_out[k][3]) a.seg_out[k][3][j] = (float)v;
      if (a.seg_all[k]) a.seg_all[k][j] = (float)v;
    }
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long u0 = (bits & 0xFFFFFFFFull) | tagw, u1 = (bits >> 32) | tagw;
    for (int p = 0; p < a.world; ++p) {
      unsigned long long* dst = a.peer_data[p] + 2 * (off + i);
      if (a.light) {
        __hip_atomic_store(dst, u0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, u1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        __hip_atomic_store(dst, u0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, u1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  // poll + sum in rank order (the same arithmetic on every rank): element i of rank r is complete when both of its units carry this
  // exchange's tag.  A dead channel (an earlier error) no longer waits: its sums are NaN
  const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  unsigned bad = 0u;
  const long long t0 = wall_clock64();
  const unsigned long long* mine = a.peer_data[a.me] + 2 * ((long)slot * a.world * a.cap);
  for (long i = first; i < a.n; i += step) {
    double acc = 0.0;
    for (int r = 0; r < a.world; ++r) {
      const unsigned long long* src = mine + 2 * ((long)r * a.cap + i);
      unsigned long long u0, u1;
      for (;;) {
        if (a.light) {
          u0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          u1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          u0 = __hip_atomic_load(src, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          u1 = __hip_atomic_load(src + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned t0u = (unsigned)(u0 >> 32), t1u = (unsigned)(u1 >> 32);
        if (t0u == tag && t1u == tag) break;
        if ((int)(t0u - tag) > 0 || (int)(t1u - tag) > 0) { bad |= 2u; break; }      // a later exchange already sits in the slot
        if (dead || bad) { bad |= dead ? 0u : 1u; break; }
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > a.timeout_ticks) { bad |= 1u; break; }
      }
      acc += __longlong_as_double((long long)((u0 & 0xFFFFFFFFull) | (u1 << 32)));
    }
    if (dead || bad) acc = __longlong_as_double(0x7FF8000000000000LL);
    if (a.nseg == 0) a.v[i] = acc;
    else {
      int k; long j;
      seg_of(a, i, k, j);
      a.seg[k][j] = acc;
      if (a.seg_zero[k])      // the consumer sums all replicas: the other slabs must not count a second time
        for (int r = 1; r < a.seg_nrep[k]; ++r) a.seg[k][(long)r * a.seg_len[k] + j] = 0.0;
    }
  }
  if (bad) __hip_atomic_fetch_or(a.err, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (t == 0) {
    const unsigned long long d = __hip_atomic_fetch_add(a.seq + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d + 1 == (unsigned long long)B) {
      __hip_atomic_store(a.seq + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.seq, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The in-kernel form of the exchange (p2p_xp.h) on a bare vector: what the fused BatchNorm kernels of bn.hip do in their prologues,
// without the BatchNorm -- for the host's acceptance test.  Workgroup 0 is the leader; element j of the result is written back by
// workgroup j % gridDim.x, i.e. almost always by a workgroup that had to wait for the leader and read the published units.
__global__ __launch_bounds__(256) void p2p_xp_test_kernel(XpArgs xp, double* v, int n) {
  XpCtx xc = xp_begin(xp);
  unsigned bad = 0u;
  xp_exchange(xp, xc, blockIdx.x == 0, 0L, n, [&](int j) { return v[j]; }, bad);
  for (int j = threadIdx.x; j < n; j += 256) {
    const double w = xp_get(xp, xc, j, bad);
    if ((unsigned)j % gridDim.x == blockIdx.x) v[j] = w;
  }
  __syncthreads();
  xp_end(xp, xc, bad, gridDim.x);
}

}  // namespace

// bytes of an IPC handle as this library hands it around (hipIpcMemHandle_t)
extern "C" int npp_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// Allocate this rank's mailboxes (channels x [SLOTS][world][cap doubles] + flags), zeroed, and write the IPC handle of the
// allocation to handle_out (npp_p2p_handle_bytes bytes).  cap_doubles: the longest vector one exchange may carry.
extern "C" int npp_p2p_alloc(int rank, int world, int64_t cap_doubles, int channels, void* handle_out) {
  NPP_REQUIRE(handle_out && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && cap_doubles > 0 && channels >= 1 &&
              channels <= P2P_MAX_CHANNELS, NPP_E_SHAPE, "npp_p2p_alloc: bad arguments (rank %d of %d, %ld doubles, %d channels)", rank,
              world, (long)cap_doubles, channels);
  NPP_REQUIRE(g.local == nullptr, NPP_E_UNSUPPORTED, "npp_p2p_alloc: mailboxes exist already (npp_p2p_close first)");
  const size_t bytes = total_bytes(cap_doubles, world, channels);
  // peer writes must not sit in a cache: uncached, else fine-grained, else plain device memory -- the first kind of allocation the
  // runtime both grants AND exports (hipIpcGetMemHandle)
  void* p = nullptr;
  hipIpcMemHandle_t h;
  hipError_t e = hipErrorUnknown;
  for (int kind = 0; kind < 3 && p == nullptr; ++kind) {
    void* q = nullptr;
    e = kind == 0 ? hipExtMallocWithFlags(&q, bytes, hipDeviceMallocUncached)
        : kind == 1 ? hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained) : hipMalloc(&q, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); continue; }
    e = hipIpcGetMemHandle(&h, q);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); continue; }
    p = q;
    g.alloc_kind = kind;
  }
  if (p == nullptr) {
    npp_set_error("npp_p2p_alloc: no exportable allocation (%s)", hipGetErrorString(e));
    return NPP_E_UNSUPPORTED;
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(p);
    npp_set_error("npp_p2p_alloc: cannot zero the mailboxes");
    return NPP_E_HIP;
  }
  void* res = nullptr;
  // per channel: [cap][2 units] results, then the XP_SUB first-level counters
  const size_t res_bytes = (size_t)channels * (cap_doubles * 2 + XP_SUB * XP_SUB_STRIDE) * sizeof(unsigned long long);
  if (hipMalloc(&res, res_bytes) != hipSuccess || hipMemset(res, 0, res_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(p);
    if (res) (void)hipFree(res);
    npp_set_error("npp_p2p_alloc: cannot allocate the result vectors");
    return NPP_E_HIP;
  }
  g.res = static_cast<unsigned long long*>(res);
  memcpy(handle_out, &h, sizeof(h));
  g.local = p; g.bytes = bytes; g.rank = rank; g.world = world; g.cap = cap_doubles; g.nchan = channels;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) g.peers[r] = nullptr;
  g.peers[rank] = p;
  map_channels(p, rank, cap_doubles, world, channels, true);
  const char* tmo = getenv("NPP_P2P_TIMEOUT_MS");
  const long long ms = tmo ? atoll(tmo) : 120000;
  g.timeout_ticks = ms * 100000LL;      // wall_clock64: 100 MHz
  return NPP_OK;
}

// Map the peers' mailboxes: handles = world x npp_p2p_handle_bytes bytes in rank order (this rank's own entry is ignored).
extern "C" int npp_p2p_open(const void* handles) {
  NPP_REQUIRE(handles && g.local, NPP_E_NULL, "npp_p2p_open: npp_p2p_alloc first");
  const char* hb = static_cast<const char*>(handles);
  for (int r = 0; r < g.world; ++r) {
    if (r == g.rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, hb + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      npp_set_error("npp_p2p_open: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      return NPP_E_UNSUPPORTED;
    }
    g.peers[r] = p;
    map_channels(p, r, g.cap, g.world, g.nchan, false);
  }
  return NPP_OK;
}

// Uncached and fine-grained mailboxes are coherent between the devices at the granularity of the 8-byte units: relaxed system-scope
// atomics carry them (one rank, tools/p2p_time.py: 3.9 us per exchange).  Plain device memory (the runtime refused both) gets
// release / acquire units instead.  NPP_P2P_LIGHT=0 forces those.
static int g_light_override = -1;      // npp_p2p_set_mode
static int p2p_light() {
  if (g_light_override >= 0) return g_light_override;
  static const int env = getenv("NPP_P2P_LIGHT") ? atoi(getenv("NPP_P2P_LIGHT")) : -1;
  if (env >= 0) return env != 0;
  return g.alloc_kind == 0 || g.alloc_kind == 1;
}
// 1: relaxed units, 0: release / acquire units, -1: the default for the allocation kind.  Returns the mode now in force.  The host's
// acceptance test (npp_amd/comm.py) tries the relaxed form first and the fenced form if ANY rank saw a wrong sum.
extern "C" int npp_p2p_set_mode(int light) { g_light_override = light < 0 ? -1 : (light != 0); return p2p_light(); }
extern "C" int npp_p2p_alloc_kind(void) { return g.local ? g.alloc_kind : -1; }
extern "C" int64_t npp_p2p_capacity(void) { return g.local ? (int64_t)g.cap : 0; }
extern "C" int npp_p2p_channels(void) { return g.local ? g.nchan : 0; }

// In-place SUM over the ranks of `count` doubles on `stream`; every rank must issue the same sequence of exchanges per channel.
extern "C" int npp_p2p_exchange(double* stats, int64_t count, int channel, void* stream) {
  NPP_REQUIRE(stats && count > 0, NPP_E_NULL, "npp_p2p_exchange: null / empty buffer");
  NPP_REQUIRE(g.local && channel >= 0 && channel < g.nchan, NPP_E_UNSUPPORTED, "npp_p2p_exchange: no mailboxes / bad channel %d", channel);
  if (count > g.cap) {
    npp_set_error("npp_p2p_exchange: %ld doubles exceed the mailbox capacity %ld (split the vector)", (long)count, g.cap);
    return NPP_E_UNSUPPORTED;
  }
  for (int r = 0; r < g.world; ++r)
    NPP_REQUIRE(g.peers[r], NPP_E_UNSUPPORTED, "npp_p2p_exchange: rank %d's mailbox is not mapped (npp_p2p_open)", r);
  const Channel& c = g.ch[channel];
  ExArgs a;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) a.peer_data[r] = r < g.world ? c.data[r] : nullptr;
  a.seq = c.seq; a.err = c.err; a.v = stats; a.n = count; a.cap = g.cap; a.me = g.rank; a.world = g.world;
  a.timeout_ticks = g.timeout_ticks;
  a.light = p2p_light();
  a.nseg = 0;
  for (int k = 0; k < P2P_MAX_SEGS; ++k) {
    a.seg[k] = nullptr; a.seg_len[k] = 0; a.seg_split[k] = 0; a.seg_nrep[k] = 0; a.seg_zero[k] = 0; a.seg_all[k] = nullptr;
    for (int q = 0; q < 4; ++q) a.seg_out[k][q] = nullptr;
  }
  int blocks = (int)((count + 2047) / 2048);      // >= 2 elements per thread before another workgroup pays
  if (blocks > P2P_MAX_BLOCKS) blocks = P2P_MAX_BLOCKS;
  hipLaunchKernelGGL(p2p_exchange_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, a);
  return npp_check_launch("p2p_exchange");
}

// Slab form: segment k is the sum over its nrep replica slabs [nrep][len] (the f64 partial sums a BatchNorm-backward reduce leaves,
// include/npp_hip.h NPP_STAT_REPLICAS); the LOCAL sums are also written as floats: elements [0, split) to out0 (and out0_dup),
// [split, 2 split) to out1, [2 split, 3 split) to out2 (elements past 3 split have no float copy) -- dbeta and dgamma of torch.nn.SyncBatchNorm, which are NOT reduced -- and the
// world's sum replaces replica 0 (zero_rest: the other replicas are zeroed, for consumers that sum all of them).
// One launch instead of npp_bn_bwd_sum + exchange: a link less in the backward chain of every SyncBatchNorm.
extern "C" int npp_p2p_exchange_slabs(const NppP2pSeg* segs, int nseg, int channel, void* stream) {
  NPP_REQUIRE(segs && nseg >= 1 && nseg <= P2P_MAX_SEGS, NPP_E_SHAPE, "npp_p2p_exchange_slabs: 1..%d segments", P2P_MAX_SEGS);
  NPP_REQUIRE(g.local && channel >= 0 && channel < g.nchan, NPP_E_UNSUPPORTED, "npp_p2p_exchange_slabs: no mailboxes / bad channel %d", channel);
  for (int r = 0; r < g.world; ++r)
    NPP_REQUIRE(g.peers[r], NPP_E_UNSUPPORTED, "npp_p2p_exchange_slabs: rank %d's mailbox is not mapped (npp_p2p_open)", r);
  const Channel& c = g.ch[channel];
  ExArgs a;
  for (int r = 0; r < P2P_MAX_WORLD; ++r) a.peer_data[r] = r < g.world ? c.data[r] : nullptr;
  a.seq = c.seq; a.err = c.err; a.v = nullptr; a.cap = g.cap; a.me = g.rank; a.world = g.world; a.timeout_ticks = g.timeout_ticks;
  a.light = p2p_light();
  a.nseg = nseg;
  long total = 0;
  for (int k = 0; k < P2P_MAX_SEGS; ++k) {
    if (k < nseg) {
      const NppP2pSeg& sg = segs[k];
      NPP_REQUIRE(sg.slabs && sg.len > 0 && sg.nrep >= 1 && sg.split >= 0 && sg.split <= sg.len, NPP_E_SHAPE, "npp_p2p_exchange_slabs: bad segment %d", k);
      a.seg[k] = sg.slabs; a.seg_len[k] = sg.len; a.seg_split[k] = sg.split; a.seg_nrep[k] = sg.nrep; a.seg_zero[k] = sg.zero_rest;
      a.seg_out[k][0] = sg.out0; a.seg_out[k][1] = sg.out1; a.seg_out[k][2] = sg.out2; a.seg_out[k][3] = sg.out0_dup;
      a.seg_all[k] = sg.out_all;
      total += sg.len;
    } else {
      a.seg[k] = nullptr; a.seg_len[k] = 0; a.seg_split[k] = 0; a.seg_nrep[k] = 0; a.seg_zero[k] = 0; a.seg_all[k] = nullptr;
      for (int q = 0; q < 4; ++q) a.seg_out[k][q] = nullptr;
    }
  }
  if (total > g.cap) {
    npp_set_error("npp_p2p_exchange_slabs: %ld doubles exceed the mailbox capacity %ld", total, g.cap);
    return NPP_E_UNSUPPORTED;
  }
  a.n = total;
  int blocks = (int)((total + 1023) / 1024);
  if (blocks > P2P_MAX_BLOCKS) blocks = P2P_MAX_BLOCKS;
  hipLaunchKernelGGL(p2p_exchange_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, a);
  return npp_check_launch("p2p_exchange_slabs");
}

// In-place SUM over the ranks of `count` doubles through the IN-KERNEL form of the exchange (the leader / follower protocol of the
// fused BatchNorm kernels, p2p_xp.h) -- for acceptance tests: the same sequence number, mailboxes and result vector a folded launch uses.
extern "C" int npp_p2p_exchange_folded_test(double* stats, int64_t count, int channel, void* stream) {
  NPP_REQUIRE(stats && count > 0, NPP_E_NULL, "npp_p2p_exchange_folded_test: null / empty buffer");
  XpArgs xp;
  const int rc = npp_p2p_xp_args(channel, (long)count, &xp);
  if (rc != NPP_OK) return rc;
  hipLaunchKernelGGL(p2p_xp_test_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, xp, stats, (int)count);
  return npp_check_launch("p2p_xp_test");
}

// 0: every exchange of every channel found its peers; otherwise the OR of the channels' error words (1: a poll timed out, 2: a
// peer overwrote a slot before this rank had read it).  Synchronises the DEVICE first (every stream, the non-blocking ones the
// exchanges run on included): an exchange still in flight has not reported yet.
extern "C" int npp_p2p_status(void) {
  if (!g.local) return 0;
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return 1; }
  unsigned bad = 0;
  for (int c = 0; c < g.nchan; ++c) {
    unsigned int e = 0;
    if (hipMemcpy(&e, g.ch[c].err, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 1; }
    bad |= e;
  }
  return (int)bad;
}

// The poll timeout (ms) of every later exchange; returns the previous value.  The host's acceptance test runs with a few seconds
// and restores the long watchdog value afterwards (npp_amd/comm.py).
extern "C" int64_t npp_p2p_set_timeout_ms(int64_t ms) {
  const long long old = g.timeout_ticks / 100000LL;
  if (ms > 0) g.timeout_ticks = (long long)ms * 100000LL;
  return (int64_t)old;
}

// Clear the channels' error words (after the host has agreed, collectively, to try the other unit mode: the exchange counters of a
// dead channel keep moving, so the ranks stay in step and a cleared channel is usable again).  Synchronises the device first.
extern "C" int npp_p2p_reset_errors(void) {
  if (!g.local) return NPP_OK;
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return NPP_E_HIP; }
  for (int c = 0; c < g.nchan; ++c)
    if (hipMemset(g.ch[c].err, 0, sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); return NPP_E_HIP; }
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return NPP_E_HIP; }
  return NPP_OK;
}

extern "C" int npp_p2p_close(void) {
  if (!g.local) return NPP_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < g.world; ++r)
    if (r != g.rank && g.peers[r]) (void)hipIpcCloseMemHandle(g.peers[r]);
  (void)hipFree(g.local);
  if (g.res) (void)hipFree(g.res);
  (void)hipGetLastError();
  g = P2P();
  return NPP_OK;
}

// The arguments of an in-kernel exchange (p2p_xp.h) of n_doubles on `channel`: the fused BatchNorm entry points of bn.hip call this
// when the caller names a channel.  The exchange takes the channel's next sequence number exactly as a launch of npp_p2p_exchange*
// on the same stream would.
int npp_p2p_xp_args(int channel, long n_doubles, XpArgs* out) {
  NPP_REQUIRE(out, NPP_E_NULL, "npp_p2p_xp_args: null");
  NPP_REQUIRE(g.local && g.res && channel >= 0 && channel < g.nchan, NPP_E_UNSUPPORTED, "in-kernel exchange: no mailboxes / bad channel %d", channel);
  if (n_doubles <= 0 || n_doubles > g.cap) {
    npp_set_error("in-kernel exchange: %ld doubles exceed the mailbox capacity %ld", n_doubles, g.cap);
    return NPP_E_UNSUPPORTED;
  }
  for (int r = 0; r < g.world; ++r)
    NPP_REQUIRE(g.peers[r], NPP_E_UNSUPPORTED, "in-kernel exchange: rank %d's mailbox is not mapped (npp_p2p_open)", r);
  const Channel& c = g.ch[channel];
  for (int r = 0; r < P2P_MAX_WORLD; ++r) out->peer_data[r] = r < g.world ? c.data[r] : nullptr;
  out->seq = c.seq; out->err = c.err; out->res = g.res + (size_t)channel * (g.cap * 2 + XP_SUB * XP_SUB_STRIDE);
  out->sub = out->res + g.cap * 2;
  out->cap = g.cap; out->timeout_ticks = g.timeout_ticks; out->me = g.rank; out->world = g.world; out->light = p2p_light(); out->pad = 0;
  return NPP_OK;
}
