// Zone-batched launches for the one-encoder-per-zone heads (EEGNet_Encoder, CVBlock, HeadConv_Paper_Version;
// fast.py:203-210: eight small encoders on eight channel subsets).  Each encoder pass is a chain of ~20 short
// kernels; eight chains of them are launch-bound however they are queued (HIP-graph kernel nodes cost ~6 us each).
//
// The host code of a head stays as it is -- one plan, one call per zone.  Between isd_zone_batch_begin() and
// isd_zone_batch_launch() every kernel launch of this thread is RECORDED instead of issued (kernel, grid, block,
// arguments); isd_zone_batch_next() closes a zone.  The launch then zips the zones' chains: launch i of every zone
// is the same kernel, so it goes out ONCE with blockIdx.z = zone and the zones' argument tuples side by side in the
// kernel argument block.  A kernel takes part through ISD_ZONE_FN + ISD_ZONE_REGISTER (its body is a __device__ function that gets
// its own zone's grid size -- and blockIdx.z / gridDim.z -- as trailing arguments: grid-stride loops must not see the
// other zones', and blockIdx.z is the zone in the multi-zone launch).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.h"

namespace isd {

constexpr int kMaxBatchZones = 8;

template <typename... P>
struct ZoneArgs {
  int n;
  unsigned gx[kMaxBatchZones], gy[kMaxBatchZones];
  std::tuple<P...> a[kMaxBatchZones];
};

#if defined(__HIPCC__)
template <typename Fn, typename... P>
__global__ __launch_bounds__(Fn::kBounds) void zone_multi_kernel(ZoneArgs<P...> m) {
  const int z = blockIdx.z;
  if (blockIdx.x >= m.gx[z] || blockIdx.y >= m.gy[z]) return;      // outside this zone's own grid
  std::apply([&](const P&... a) { Fn::call(a..., m.gx[z], m.gy[z], 0u, 1u); }, m.a[z]);
}

struct ZoneFill {
  int n;
  void* p[kMaxBatchZones];
  unsigned long long n4[kMaxBatchZones];                            // 4-byte words to clear
};
__global__ __launch_bounds__(256) inline void zone_fill_kernel(ZoneFill m) {
  const int z = blockIdx.y;
  unsigned* d = reinterpret_cast<unsigned*>(m.p[z]);
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < m.n4[z]; i += (unsigned long long)gridDim.x * 256)
    d[i] = 0u;
}
#endif

// one recorded operation of one zone
struct ZoneOp {
  // kind 0: kernel; 1: clear
  int kind = 0;
  const void* kernel = nullptr;                                     // the __global__ stub's address: the chain's identity
  // zips `n` recorded argument tuples (type-erased) into one launch
  hipError_t (*zip)(int n, const ZoneOp* const* ops, hipStream_t st) = nullptr;
  dim3 grid, block;
  unsigned lds = 0;
  std::shared_ptr<void> args;                                       // std::tuple<P...>
  void* ptr = nullptr;                                              // clear
  size_t bytes = 0;
};

struct ZoneRecorder {
  bool active = false;
  std::vector<std::vector<ZoneOp>> zones;
};
ZoneRecorder& zone_recorder();                                      // thread-local (zonebatch.hip)

#if defined(__HIPCC__)
template <typename Fn, typename... P>
hipError_t zone_zip(int n, const ZoneOp* const* ops, hipStream_t st) {
  ZoneArgs<P...> m;
  m.n = n;
  unsigned gx = 1, gy = 1, lds = 0;
  for (int z = 0; z < n; ++z) {
    m.gx[z] = ops[z]->grid.x;
    m.gy[z] = ops[z]->grid.y;
    m.a[z] = *static_cast<const std::tuple<P...>*>(ops[z]->args.get());
    gx = m.gx[z] > gx ? m.gx[z] : gx;
    gy = m.gy[z] > gy ? m.gy[z] : gy;
    lds = ops[z]->lds > lds ? ops[z]->lds : lds;
  }
  for (int z = n; z < kMaxBatchZones; ++z) {
    m.gx[z] = m.gy[z] = 0;
    m.a[z] = m.a[0];
  }
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)zone_multi_kernel<Fn, P...>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((zone_multi_kernel<Fn, P...>), dim3(gx, gy, (unsigned)n), ops[0]->block, lds, st, m);
  return hipGetLastError();
}

// registry: kernel stub -> zip function (filled by the static registrars of ISD_ZONE_REGISTER)
std::unordered_map<const void*, hipError_t (*)(int, const ZoneOp* const*, hipStream_t)>& zone_registry();
struct ZoneRegistrar {
  ZoneRegistrar(const void* k, hipError_t (*zip)(int, const ZoneOp* const*, hipStream_t)) { zone_registry()[k] = zip; }
};

// Launch now, or record when a zone batch is open on this thread.  Returns false when the kernel cannot be recorded.
template <typename... P, typename... A>
bool zone_launch(void (*k)(P...), dim3 grid, dim3 block, unsigned lds, hipStream_t st, A&&... args) {
  static_assert(sizeof...(P) == sizeof...(A), "argument count");
  ZoneRecorder& r = zone_recorder();
  if (!r.active) {
    hipLaunchKernelGGL(k, grid, block, lds, st, static_cast<P>(args)...);
    return true;
  }
  auto it = zone_registry().find(reinterpret_cast<const void*>(k));
  if (it == zone_registry().end() || grid.z != 1) {
    set_error("zone batch: a kernel of this call is not zone-batchable");
    r.active = false;                                               // poison: isd_zone_batch_launch reports it
    r.zones.clear();
    return false;
  }
  ZoneOp op;
  op.kind = 0;
  op.kernel = reinterpret_cast<const void*>(k);
  op.zip = it->second;
  op.grid = grid;
  op.block = block;
  op.lds = lds;
  op.args = std::make_shared<std::tuple<P...>>(static_cast<P>(args)...);
  r.zones.back().push_back(std::move(op));
  return true;
}

// hipMemsetAsync(ptr, 0, bytes) -- or its record (ptr and bytes must be multiples of 4)
inline hipError_t zone_clear(void* ptr, size_t bytes, hipStream_t st) {
  ZoneRecorder& r = zone_recorder();
  if (!r.active) return hipMemsetAsync(ptr, 0, bytes, st);
  ZoneOp op;
  op.kind = 1;
  op.ptr = ptr;
  op.bytes = bytes;
  r.zones.back().push_back(std::move(op));
  return hipSuccess;
}

// A zone-batchable kernel NAME: `NAME_body(params..., zgx, zgy, zbz, zgz)` is the __device__ body (zgx / zgy / zgz
// stand for gridDim.x / .y / .z of the zone's own launch, zbz for its blockIdx.z).  ISD_ZONE_FN defines the functor the
// multi-zone kernel calls, ISD_ZONE_REGISTER ties it to the plain __global__ entry point (kernel templates:
// ISD_ZONE_FN_T, then one ISD_ZONE_REGISTER_T per instantiation that is launched).
#define ISD_ZONE_FN(NAME, BOUNDS)                                                        \
  struct NAME##_zfn {                                                                    \
    static constexpr int kBounds = BOUNDS;                                               \
    template <typename... A>                                                             \
    __device__ __forceinline__ static void call(A... a) { NAME##_body(a...); }           \
  };
template <typename Fn, typename... P>
auto zone_zip_of(void (*)(P...)) -> hipError_t (*)(int, const ZoneOp* const*, hipStream_t) { return &zone_zip<Fn, P...>; }
#define ISD_ZONE_REGISTER(NAME) \
  static ZoneRegistrar NAME##_zreg(reinterpret_cast<const void*>(&NAME), zone_zip_of<NAME##_zfn>(&NAME));
// kernel templates: functor template via ISD_ZONE_FN_T, one registration per launched instantiation
#define ISD_ZONE_FN_T(NAME, BOUNDS, TPARAM)                                              \
  template <TPARAM V>                                                                    \
  struct NAME##_zfn {                                                                    \
    static constexpr int kBounds = BOUNDS;                                               \
    template <typename... A>                                                             \
    __device__ __forceinline__ static void call(A... a) { NAME##_body<V>(a...); }        \
  };
#define ISD_ZONE_REGISTER_T(NAME, V) \
  static ZoneRegistrar NAME##_zreg_##V(reinterpret_cast<const void*>(&NAME<V>), zone_zip_of<NAME##_zfn<V>>(&NAME<V>));
#endif  // __HIPCC__

}  // namespace isd
