// Recorder, registry and C ABI of the zone-batched launches (zonebatch.h).
#include "zonebatch.h"

#include "../../include/isd_hip.h"

namespace isd {

ZoneRecorder& zone_recorder() {
  static thread_local ZoneRecorder r;
  return r;
}

std::unordered_map<const void*, hipError_t (*)(int, const ZoneOp* const*, hipStream_t)>& zone_registry() {
  static std::unordered_map<const void*, hipError_t (*)(int, const ZoneOp* const*, hipStream_t)> m;
  return m;
}

}  // namespace isd

using namespace isd;

extern "C" int isd_zone_batch_begin(void) {
  ZoneRecorder& r = zone_recorder();
  ISD_CHECK_ARG(!r.active, "isd_zone_batch_begin: a zone batch is already open on this thread");
  r.zones.clear();
  r.zones.emplace_back();
  r.active = true;
  return ISD_OK;
}

extern "C" int isd_zone_batch_next(void) {
  ZoneRecorder& r = zone_recorder();
  ISD_CHECK_ARG(r.active, "isd_zone_batch_next: no zone batch is open (or a call in it could not be recorded)");
  ISD_CHECK_ARG((int)r.zones.size() < kMaxBatchZones, "isd_zone_batch_next: more than %d zones", kMaxBatchZones);
  r.zones.emplace_back();
  return ISD_OK;
}

extern "C" int isd_zone_batch_abort(void) {
  ZoneRecorder& r = zone_recorder();
  r.active = false;
  r.zones.clear();
  return ISD_OK;
}

extern "C" int isd_zone_batch_launch(void* stream) {
  ZoneRecorder& r = zone_recorder();
  if (!r.active) {
    r.zones.clear();
    set_error("isd_zone_batch_launch: no zone batch is open (or a call in it could not be recorded)");
    return ISD_ERR_INVALID;
  }
  r.active = false;
  std::vector<std::vector<ZoneOp>> zones;
  zones.swap(r.zones);
  while (!zones.empty() && zones.back().empty()) zones.pop_back();        // a trailing isd_zone_batch_next
  const int n = (int)zones.size();
  if (n == 0) return ISD_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t len = zones[0].size();
  for (int z = 1; z < n; ++z)
    ISD_CHECK_ARG(zones[z].size() == len, "isd_zone_batch_launch: zone %d recorded %zu launches, zone 0 %zu", z,
                  zones[z].size(), len);
  const ZoneOp* ops[kMaxBatchZones];
  // every chain is validated before the first launch goes out: a mismatch must not leave half a stage issued
  for (size_t i = 0; i < len; ++i)
    for (int z = 1; z < n; ++z) {
      const ZoneOp &o = zones[z][i], &o0 = zones[0][i];
      ISD_CHECK_ARG(o.kind == o0.kind && o.kernel == o0.kernel && o.block.x == o0.block.x && o.block.y == o0.block.y,
                    "isd_zone_batch_launch: launch %zu differs between zone 0 and zone %d", i, z);
    }
  for (size_t i = 0; i < len; ++i) {
    for (int z = 0; z < n; ++z) ops[z] = &zones[z][i];
    if (ops[0]->kind == 0) {
      ISD_HIP_TRY(ops[0]->zip(n, ops, st));
    } else {
      ZoneFill f;
      f.n = n;
      size_t most = 0;
      for (int z = 0; z < kMaxBatchZones; ++z) {
        const ZoneOp* o = ops[z < n ? z : 0];
        ISD_CHECK_ARG((((uintptr_t)o->ptr | o->bytes) & 3) == 0, "isd_zone_batch_launch: unaligned clear");
        f.p[z] = o->ptr;
        f.n4[z] = z < n ? o->bytes / 4 : 0;
        most = f.n4[z] > most ? f.n4[z] : most;
      }
      const unsigned gx = (unsigned)(most / 256 + 1 < 64 ? most / 256 + 1 : 64);
      hipLaunchKernelGGL(zone_fill_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, f);
      ISD_LAUNCH_CHECK();
    }
  }
  return ISD_OK;
}
