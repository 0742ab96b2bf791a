// Grouped launches: K engine contexts of the SAME shape stepped as ONE sequence of launches (eae_group_train_step).
//
// The reference's real workload is a grid of independent configurations at batch 64 (R.md:246, 599-711): one such step is ~57
// dependent launches of kernels that fill an eighth of the GPU, and K contexts on K streams are bound by the dispatch rate (~2.3 us
// per kernel whatever K is: bench.py grid_b64, 0.47x of the B=512 rate).  Here every kernel of the step has a grouped twin
//
//     __global__ k_g(GroupPack<Args> p, int gz)      grid (gx, gy, gz * n): workgroup (x, y, z) works for member z / gz with
//                                                    the member's OWN argument block p.a[z / gz] and blockIdx.z' = z % gz
//
// that runs the unchanged kernel body, so a member's arithmetic is instruction for instruction what it is alone (bitwise equal, tested).
// The host side needs no second copy of the step logic either: the step functions run once per member with a RECORDER installed --
// every launch, event operation and copy is appended to the member's list instead of being enqueued -- and the K lists are then
// zipped: position i of every list must be the same kernel with the same grid on the same stream slot, and is enqueued once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

constexpr int EAE_GROUP_MAX = 8;          // members per grouped launch at most (a larger group is enqueued in several launches)
constexpr int EAE_GROUP_KERNARG = 3840;   // bytes of kernel arguments a grouped launch may carry (HIP's limit is 4 KB incl. hidden ones)

template <class A> struct GroupPack {
  static constexpr int CAP = (int)(EAE_GROUP_KERNARG / sizeof(A)) < EAE_GROUP_MAX ? (int)(EAE_GROUP_KERNARG / sizeof(A)) : EAE_GROUP_MAX;
  static_assert(CAP == EAE_GROUP_MAX, "argument block too large: a group of 8 would need two launches (6-7 us each on the dependency chain)");
  A a[CAP];
};

#if defined(__HIPCC__)
// the member's argument block, read from the kernel-argument segment with scalar loads (the pack is the FIRST kernel parameter; a
// by-value parameter indexed with a run-time value would be copied to scratch first)
template <class A> __device__ __forceinline__ A group_args(int gz) {
  A a;
  const char* base = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  __builtin_memcpy(&a, base + (size_t)(blockIdx.z / (unsigned)gz) * sizeof(A), sizeof(A));
  return a;
}
// XCD-aware kernels (their workgroups that share operands must share an L2): the hardware hands the linear workgroup id
// L = blockIdx.z * gridDim.x + blockIdx.x round-robin to the 8 XCDs, so a member's few workgroups land on eight different ones and every
// operand tile is fetched once per XCD (measured: the grouped weight gradients moved 3.2-5x their algorithmic bytes, 1.15-1.8x alone).
// Here every XCD takes a CONTIGUOUS run of the group's logical ids (member-major): adjacent (slice, block) pairs of one member sit
// on one XCD.  Returns the member and the workgroup's logical id inside the member (gz == 1 kernels only).
__device__ __forceinline__ void group_xcd_map(unsigned& member, int& li) {
  const unsigned gx = gridDim.x, T = gx * gridDim.z, L = blockIdx.z * gx + blockIdx.x;
  const unsigned g = (T & 7u) == 0 ? (L & 7u) * (T >> 3) + (L >> 3) : L;
  member = g / gx; li = (int)(g % gx);
}
template <class A> __device__ __forceinline__ A group_args_of(unsigned member) {
  A a;
  const char* base = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  __builtin_memcpy(&a, base + (size_t)member * sizeof(A), sizeof(A));
  return a;
}
#endif

typedef int (*GroupLaunchFn)(const void* kg, dim3 grid, dim3 block, unsigned smem, hipStream_t st, const unsigned char* const* args, int n);

struct GroupItem {
  enum Kind : uint8_t { LAUNCH, EV_RECORD, EV_WAIT, OP } kind;
  int8_t slot;                 // 0 = the caller's stream, 1 + k = side stream k of the context
  const void* kg;              // LAUNCH: the grouped twin
  GroupLaunchFn fn;
  dim3 grid, block;
  unsigned smem;
  uint32_t arg_off, arg_size;  // the member's argument block inside GroupRec::argbuf
  hipEvent_t ev;               // EV_RECORD / EV_WAIT
  std::function<int(hipStream_t)> op;   // OP: anything else (a memset, a copy): executed per member on the slot's stream
};

struct GroupRec {
  void* ctx = nullptr;                     // eae_ctx of the member
  hipStream_t user = nullptr;              // the stream the member's step was called with
  int (*slot_of)(void* ctx, hipStream_t user, hipStream_t st) = nullptr;
  int error = 0;
  std::string msg;                         // the failing member's message (recorded on a worker thread: eae_last_error is thread-local)
  std::vector<GroupItem> items;
  std::vector<unsigned char> argbuf;
  void clear() { items.clear(); argbuf.clear(); error = 0; }
};

extern thread_local GroupRec* eae_rec;     // non-null: launches made by this thread are recorded (eae_api.hip)
// Launchers that pick a tile geometry by the size of the grid (eae_conv_launch.hip) see batch x eae_geo_mult: the members of a grouped
// step run as ONE launch.  Thread-local, 1 outside eae_group_train_step; eae_set_geometry_mult() sets it for a context run alone
// (the bitwise group-vs-alone test).
extern thread_local int eae_geo_mult;

int eae_rec_fail(const char* what);        // marks the recording as unusable (a launch with no grouped twin, an unknown stream)
inline GroupItem* eae_rec_add(GroupItem::Kind kind, hipStream_t st) {
  GroupRec* r = eae_rec;
  const int slot = r->slot_of(r->ctx, r->user, st);
  if (slot < 0) { eae_rec_fail("grouped step: work on a stream that is not one of the context's"); return nullptr; }
  r->items.emplace_back();
  GroupItem& it = r->items.back();
  it.kind = kind; it.slot = (int8_t)slot; it.kg = nullptr; it.fn = nullptr; it.smem = 0; it.arg_off = it.arg_size = 0; it.ev = nullptr;
  return &it;
}

template <class A> int group_launch(const void* kg, dim3 grid, dim3 block, unsigned smem, hipStream_t st, const unsigned char* const* args, int n) {
  const unsigned gz = grid.z;
  for (int k0 = 0; k0 < n; k0 += GroupPack<A>::CAP) {
    GroupPack<A> p;
    const int m = n - k0 < GroupPack<A>::CAP ? n - k0 : GroupPack<A>::CAP;
    for (int i = 0; i < m; ++i) memcpy(static_cast<void*>(&p.a[i]), args[k0 + i], sizeof(A));
    grid.z = gz * (unsigned)m;
    hipLaunchKernelGGL(reinterpret_cast<void (*)(GroupPack<A>, int)>(const_cast<void*>(kg)), grid, block, smem, st, p, (int)gz);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Every launch of the step goes through here: `k` runs now, or its grouped twin `kg` is recorded.
template <class A>
inline void eae_launch(void (*k)(A), void (*kg)(GroupPack<A>, int), dim3 grid, dim3 block, unsigned smem, hipStream_t st, const A& a) {
  if (eae_rec) {
    if (GroupItem* it = eae_rec_add(GroupItem::LAUNCH, st)) {
      it->kg = reinterpret_cast<const void*>(kg); it->fn = &group_launch<A>;
      it->grid = grid; it->block = block; it->smem = smem;
      std::vector<unsigned char>& buf = eae_rec->argbuf;
      it->arg_off = (uint32_t)buf.size(); it->arg_size = (uint32_t)sizeof(A);
      buf.resize(buf.size() + sizeof(A));
      memcpy(buf.data() + it->arg_off, static_cast<const void*>(&a), sizeof(A));
    }
    return;
  }
  hipLaunchKernelGGL(k, grid, block, smem, st, a);
}
// a launch that has no grouped twin: fine outside a recording, an error inside one
#define EAE_NO_GROUP(what) do { if (eae_rec) return eae_rec_fail("grouped step: " what " has no grouped form"); } while (0)

// event operations and copies of the step path
inline hipError_t eae_event_record(hipEvent_t ev, hipStream_t st) {
  if (eae_rec) { if (GroupItem* it = eae_rec_add(GroupItem::EV_RECORD, st)) it->ev = ev; return hipSuccess; }
  return hipEventRecord(ev, st);
}
inline hipError_t eae_stream_wait_event(hipStream_t st, hipEvent_t ev) {
  if (eae_rec) { if (GroupItem* it = eae_rec_add(GroupItem::EV_WAIT, st)) it->ev = ev; return hipSuccess; }
  return hipStreamWaitEvent(st, ev, 0);
}
inline hipError_t eae_memset_async(void* p, int v, size_t bytes, hipStream_t st) {
  if (eae_rec) {
    if (GroupItem* it = eae_rec_add(GroupItem::OP, st)) it->op = [=](hipStream_t s) { return hipMemsetAsync(p, v, bytes, s) == hipSuccess ? 0 : -3; };
    return hipSuccess;
  }
  return hipMemsetAsync(p, v, bytes, st);
}
inline hipError_t eae_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (eae_rec) {
    if (GroupItem* it = eae_rec_add(GroupItem::OP, st)) it->op = [=](hipStream_t s) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -3; };
    return hipSuccess;
  }
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
}
