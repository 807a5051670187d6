// In-kernel form of the SyncBatchNorm statistics exchange of csrc/p2p.hip ("xp": exchange in the prologue).
//
// The fused BatchNorm kernels (npp_affine_add_fin*, npp_bn_bwd_apply*_fin, their multi-job forms) already start with a prologue in
// which every workgroup collapses the statistics' replica slabs into per-channel sums.  Under SyncBatchNorm those sums used to be
// exchanged by a launch of their own in front of the kernel (p2p_exchange_kernel: ~450 links of 5-8 us in the dependent chain of a
// step).  Here the LEADER workgroup of the kernel (blockIdx.x == 0 of a job) pushes its local sums into every rank's mailbox, polls
// its own mailbox, sums the world's values in rank order (the same arithmetic as the stand-alone kernel: bit-identical on every
// rank) and publishes them as tagged units in a local result vector; every other workgroup waits until the vector's last element
// carries this exchange's tag and reads its channels' sums from the vector instead of summing replicas.  Same wire format, same
// mailboxes, same sequence counter as the stand-alone kernel, so both forms can follow each other on one channel: an exchange is
// still "the s-th of this channel" and carries tag s + 1.
//
//   * every workgroup reads the channel's counter s when it starts; the counter moves to s + 1 when the LAST workgroup of the grid
//     has passed its prologue (xp_end) -- every workgroup has read s by then, and the leader has finished its polls;
//   * the result vector is one slot: the next exchange of the channel is a later kernel on the same stream;
//   * no lane ever waits for another lane of its own wave: the waits are a poll of OTHER ranks' stores (leader) or of the
//     leader's last published element (one thread per workgroup, then a workgroup barrier); a unit read before its tag arrived is
//     read again;
//   * the leader of a job is dispatched before the job's other workgroups, so the workgroups waiting for it cannot starve it;
//   * a launch that carries an exchange has at most XP_MAX_BLOCKS = 192 workgroups of 256 threads over all its jobs: its waiting
//     workgroups wait for OTHER ranks, so what they occupy must never be what another rank's progress needs.  (a) Were they allowed
//     to hold every workgroup slot, the kernel of the other branch stream could not start its leader.  (b) Even two per CU are too
//     many: a kernel that needs a CU's whole register file (conv_g8: 512 threads x ~250 VGPRs) cannot start on a CU that holds ONE
//     waiting workgroup -- rank X waiting in stream A's kernel with conv_g8 pending on stream B, rank Y waiting in stream B's kernel
//     with conv_g8 pending on stream A, is a cycle through both ranks.  192 workgroups leave >= 64 CUs free of this launch: the
//     pending kernel makes progress there, whatever the other stream is waiting for (a rank has at most one such launch per stream,
//     and in the cycle only ONE of them is waiting).  The stand-alone exchange kernel never had the problem: <= 8 workgroups.
//     Cost of 192 against 512 in the 1-rank rehearsal: 39.2 -> 39.6 ms (tools/r5_xp_budget.sh).
// Errors as in p2p.hip: a poll that times out or finds a slot overwritten sets the channel's error word and yields NaN sums.
#pragma once
#include "common.h"

constexpr int P2P_SLOTS = 4;
constexpr int P2P_MAX_WORLD = 16;
constexpr int XP_SUB = 16;             // first-level counters of the "every workgroup is past its prologue" count
constexpr int XP_SUB_STRIDE = 32;      // ... 256 bytes apart (in 8-byte words)
constexpr int XP_KEEP = 4;             // elements per leader thread kept in registers instead of travelling through the own mailbox
constexpr int XP_MAX_BLOCKS = 192;     // workgroups of a launch that carries an exchange (3/4 of the 256 CUs at one each), see above

struct XpArgs {
  unsigned long long* peer_data[P2P_MAX_WORLD];      // this channel's mailbox on every rank, as mapped into this process
  unsigned long long* seq;       // [0] exchange counter, [1] workgroups of the current exchange that are done
  unsigned int* err;
  unsigned long long* res;       // local result vector: [cap][2 units {data32 | tag32}]
  unsigned long long* sub;       // local: [XP_SUB] counters, XP_SUB_STRIDE words apart
  long cap;
  long long timeout_ticks;
  int me, world;                 // world == 0: no exchange (the kernel sums its local replicas as before)
  int light, pad;
};

// Host: the arguments of an in-kernel exchange of n doubles on `channel` (NPP_OK), or why not.  Defined in p2p.hip.
int npp_p2p_xp_args(int channel, long n_doubles, XpArgs* out);
inline void npp_xp_off(XpArgs* x) { memset(x, 0, sizeof(*x)); }

struct XpCtx {
  unsigned long long s, tagw;
  unsigned tag;
  long off_me;                          // [slot][me] of a mailbox, in doubles
  const unsigned long long* mine;       // [slot] of this rank's mailbox
  long long t0;
  bool dead;
};

NPP_DEV XpCtx xp_begin(const XpArgs& x) {
  XpCtx c;
  c.s = __hip_atomic_load(x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  c.tag = (unsigned)(c.s + 1);
  c.tagw = (unsigned long long)c.tag << 32;
  const int slot = (int)(c.s % P2P_SLOTS);
  c.off_me = ((long)slot * x.world + x.me) * x.cap;
  c.mine = x.peer_data[x.me] + 2 * ((long)slot * x.world * x.cap);
  c.t0 = wall_clock64();
  c.dead = __hip_atomic_load(x.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  return c;
}

// The whole exchange of one job, called by EVERY thread of EVERY workgroup of the job before the prologue's coefficient loop
// (contains __syncthreads()).  Leader workgroup: element j of the job's n elements is local(j) on this rank; every value is pushed
// to all mailboxes first, then the world's values are polled and summed in rank order (one round trip for the whole vector, as in
// the stand-alone kernel) and published in the result vector as tagged units.  Every other workgroup: ONE thread polls the units of
// the vector's last element (~1000 workgroups polling all of their elements saturated the L2 channel that holds them; a flag behind a
// device-scope release / acquire pair wrote back / invalidated the XCD's whole L2: +50 us per exchange), then everybody reads its
// elements with xp_get, which checks the tags and re-reads the rare unit that the hint overtook.
template <typename Local>
NPP_DEV void xp_exchange(const XpArgs& x, const XpCtx& c, bool leader, long xoff, int n, Local local, unsigned& bad) {
  const int t = threadIdx.x, nt = blockDim.x;
  if (leader) {
    // this rank's own values stay in registers (XP_KEEP elements per thread: vectors up to XP_KEEP * 256 doubles, every BatchNorm
    // of the networks; longer ones go through this rank's own mailbox like a peer's): nothing is sent to oneself, the sum below
    // still adds the ranks' values in rank order
    double own[XP_KEEP];
    const bool keep = n <= XP_KEEP * nt;
#pragma unroll
    for (int q = 0; q < XP_KEEP; ++q) {
      const int j = t + q * nt;
      if (!keep || j >= n) break;
      own[q] = local(j);
      const unsigned long long bits = (unsigned long long)__double_as_longlong(own[q]);
      const unsigned long long u0 = (bits & 0xFFFFFFFFull) | c.tagw, u1 = (bits >> 32) | c.tagw;
      for (int p = 0; p < x.world; ++p) {
        if (p == x.me) continue;
        unsigned long long* dst = x.peer_data[p] + 2 * (c.off_me + xoff + j);
        if (x.light) {
          __hip_atomic_store(dst, u0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dst + 1, u1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          __hip_atomic_store(dst, u0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dst + 1, u1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    for (int j = keep ? n : t; j < n; j += nt) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(local(j));
      const unsigned long long u0 = (bits & 0xFFFFFFFFull) | c.tagw, u1 = (bits >> 32) | c.tagw;
      for (int p = 0; p < x.world; ++p) {
        unsigned long long* dst = x.peer_data[p] + 2 * (c.off_me + xoff + j);
        if (x.light) {
          __hip_atomic_store(dst, u0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dst + 1, u1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          __hip_atomic_store(dst, u0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dst + 1, u1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    for (int j = t, q = 0; j < n; j += nt, ++q) {
      double acc = 0.0;
      for (int r = 0; r < x.world; ++r) {
        if (keep && r == x.me) {      // (XP_KEEP is small and the loop over q is not unrolled: pick the register by comparison)
          double mine_v = own[0];
#pragma unroll
          for (int k = 1; k < XP_KEEP; ++k) mine_v = q == k ? own[k] : mine_v;
          acc += mine_v;
          continue;
        }
        const unsigned long long* src = c.mine + 2 * ((long)r * x.cap + xoff + j);
        unsigned long long w0, w1;
        for (;;) {
          if (x.light) {
            w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          } else {
            w0 = __hip_atomic_load(src, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            w1 = __hip_atomic_load(src + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          const unsigned t0u = (unsigned)(w0 >> 32), t1u = (unsigned)(w1 >> 32);
          if (t0u == c.tag && t1u == c.tag) break;
          if ((int)(t0u - c.tag) > 0 || (int)(t1u - c.tag) > 0) { bad |= 2u; break; }      // a later exchange already sits in the slot
          if (c.dead || bad) { bad |= c.dead ? 0u : 1u; break; }
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - c.t0 > x.timeout_ticks) { bad |= 1u; break; }
        }
        acc += __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
      }
      if (c.dead || bad) acc = __longlong_as_double(0x7FF8000000000000LL);
      // published as two tagged units, like the mailbox's: a reader that sees this exchange's tag in a unit holds its data bits
      const unsigned long long rb = (unsigned long long)__double_as_longlong(acc);
      __hip_atomic_store(x.res + 2 * (xoff + j), (rb & 0xFFFFFFFFull) | c.tagw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(x.res + 2 * (xoff + j) + 1, (rb >> 32) | c.tagw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (no flag, no fence: the tags validate every unit on their own; the other workgroups watch the LAST element's units as a hint)
  } else {
    if (t == 0) {
      const unsigned long long* hint = x.res + 2 * (xoff + n - 1);
      for (;;) {
        const unsigned long long w0 = __hip_atomic_load(hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = __hip_atomic_load(hint + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w0 >> 32) == c.tag && (unsigned)(w1 >> 32) == c.tag) break;
        __builtin_amdgcn_s_sleep(4);
        // (the leader always publishes, NaN after its own time-out: twice its allowance before giving up on it)
        if (wall_clock64() - c.t0 > 2 * x.timeout_ticks) { bad |= 1u; break; }
      }
    }
    __syncthreads();
  }
}

// the world's element e (after xp_exchange): normally one pass -- the last element's units were there; a unit whose tag is still the
// old one (the hint overtook it) is simply read again
NPP_DEV double xp_get(const XpArgs& x, const XpCtx& c, long e, unsigned& bad) {
  const unsigned long long* src = x.res + 2 * e;
  unsigned long long w0, w1;
  for (;;) {
    w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(w0 >> 32) == c.tag && (unsigned)(w1 >> 32) == c.tag) break;
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - c.t0 > 2 * x.timeout_ticks) { bad |= 1u; return __longlong_as_double(0x7FF8000000000000LL); }
  }
  return __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
}

// after the __syncthreads() that ends the prologue, by every thread of every workgroup of the grid (nblocks of them).  The workgroups
// are counted in two levels -- XP_SUB counters 256 bytes apart, the workgroup that completes one of them bumps the channel's -- because
// ~1000 device-scope atomics on ONE address drain one after the other (measured: +13 us per kernel, more than the launch saved).
NPP_DEV void xp_end(const XpArgs& x, const XpCtx& c, unsigned bad, unsigned nblocks) {
  if (bad) __hip_atomic_fetch_or(x.err, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned k = lin % XP_SUB;
    const unsigned mine = (nblocks - k + XP_SUB - 1) / XP_SUB;      // workgroups with this residue
    unsigned long long* sub = x.sub + (size_t)k * XP_SUB_STRIDE;
    const unsigned long long d = __hip_atomic_fetch_add(sub, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d + 1 == (unsigned long long)mine) {
      __hip_atomic_store(sub, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned groups = nblocks < (unsigned)XP_SUB ? nblocks : (unsigned)XP_SUB;
      const unsigned long long e = __hip_atomic_fetch_add(x.seq + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e + 1 == (unsigned long long)groups) {
        __hip_atomic_store(x.seq + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(x.seq, c.s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read by the NEXT kernel of the stream)
      }
    }
  }
}
