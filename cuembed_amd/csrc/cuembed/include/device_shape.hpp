// What the launch heuristics need to know about the GPU they launch on.
//
// The reference asks the runtime at every EmbeddingBackward call (cudaGetDevice + two attribute queries,
// embedding_lookup.cuh:355-363).  Here the attributes are read ONCE per device id (a small table indexed by the
// id; the first caller claims the slot with a compare-and-swap, writes it and publishes it, a caller that races with
// it answers from its own query instead of reading a half-written slot) and every call only pays hipGetDevice.  A full MI355X reports 256 compute units in 8 XCDs with 4 MiB of L2 each; a CPX / DPX / QPX
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
  // 0 = empty, 1 = one thread is writing `shape`, 2 = published.  Only the thread that moved 0 -> 1 writes the
  // (non-atomic) struct; nobody reads it before the release store of 2.
  if (slot.ready.load(std::memory_order_acquire) == 2) return slot.shape;
  const DeviceShape mine = QueryDeviceShape(device);
  int expected = 0;
  if (slot.ready.compare_exchange_strong(expected, 1, std::memory_order_acq_rel)) {
    slot.shape = mine;
    slot.ready.store(2, std::memory_order_release);
  }
  return mine;   // (the writer and every racing first caller: the same values, from their own query)
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_DEVICE_SHAPE_HPP_
