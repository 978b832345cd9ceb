// What the launch heuristics need to know about the GPU they launch on.
//
// The reference asks the runtime at every EmbeddingBackward call (cudaGetDevice + two attribute queries,
// embedding_lookup.cuh:355-363).  Here the attributes are read ONCE per device id (a small table indexed by the
// id, filled on first use: no lock, the values are the same whoever writes them) and every call only pays
// hipGetDevice.  A full MI355X reports 256 compute units in 8 XCDs with 4 MiB of L2 each; a CPX / DPX / QPX
// partition reports its own share, and every XCD-aware mapping (column slices of the backward gather, tile maps of
// the radix sort) follows what is reported instead of assuming the whole chip.
#ifndef CUEMBED_INCLUDE_DEVICE_SHAPE_HPP_
#define CUEMBED_INCLUDE_DEVICE_SHAPE_HPP_

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>

namespace cuembed {
namespace detail {

struct DeviceShape {
  int compute_units;        //!< CUs of the (possibly partitioned) device
  int xcds;                 //!< XCDs = private L2s; workgroup b of a 1-D grid runs on XCD b % xcds
  int lanes_per_cu;         //!< resident lanes per CU (2048 on CDNA)
  size_t l2_bytes_per_xcd;  //!< one XCD's L2
};

//! The full MI355X (used when no device can be asked: host-only planning, tests).
inline DeviceShape Mi355xShape() { return DeviceShape{256, 8, 2048, size_t{4} << 20}; }

inline DeviceShape QueryDeviceShape(const int device) {
  DeviceShape s = Mi355xShape();
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) s.compute_units = v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && v > 0) s.xcds = v;
  else (void)hipGetLastError();   // (an older runtime without the attribute: keep 8, clear the sticky error)
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxThreadsPerMultiProcessor, device) == hipSuccess && v > 0)
    s.lanes_per_cu = v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, device) == hipSuccess && v > 0)
    s.l2_bytes_per_xcd = static_cast<size_t>(v);
  if (s.xcds > s.compute_units) s.xcds = 1;
  return s;
}

//! Shape of the CURRENT device, cached per device id.
inline DeviceShape CurrentDeviceShape() {
  constexpr int kMaxDevices = 64;
  struct Slot {
    std::atomic<int> ready{0};
    DeviceShape shape{};
  };
  static Slot slots[kMaxDevices];
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) return Mi355xShape();
  Slot& slot = slots[device];
  if (slot.ready.load(std::memory_order_acquire) == 0) {
    slot.shape = QueryDeviceShape(device);   // (racing first callers write the same values)
    slot.ready.store(1, std::memory_order_release);
  }
  return slot.shape;
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_DEVICE_SHAPE_HPP_
