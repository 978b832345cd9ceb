// Tuning harness (not part of the product): the C2-type forward kernel
// (fp16 table, fp32 accumulate, int32 indices, 16 B per lane, fixed hotness staged in LDS)
// instantiated with different (unroll, pipelined, launch-bound) settings, so that
// tools/tune_forward.py can time them side by side in ONE process.
#include <hip/hip_runtime.h>
#include "cuembed/include/embedding_lookup.hpp"

using namespace cuembed::detail;

template <int U, bool P, int BT, int SLICES = 1>
static void Launch(const void* table, int width, int batch, const int* indices, int num_hots,
                   void* out, int samples_per_block, hipStream_t stream) {
  const int lanes = width / 8 / SLICES;
  samples_per_block *= SLICES;   // keep the workgroup at the same number of threads
  // one workgroup per (sample group, slice); sample groups padded to a multiple of 8 / SLICES
  const int groups = (batch + samples_per_block - 1) / samples_per_block;
  const int per_slice = 8 / (SLICES > 1 ? SLICES : 8);
  const int rounds = SLICES > 1 ? (groups + per_slice - 1) / per_slice : 0;
  const dim3 block(lanes, samples_per_block, 1);
  const dim3 grid(SLICES > 1 ? rounds * 8 : groups, 1, 1);
  const size_t lds = static_cast<size_t>(samples_per_block) * num_hots * sizeof(int);
  GatherReduceKernel<_Float16, float, int, int, 8, false, IndexSource::kLdsStaged, U, P, BT>
      <<<grid, block, lds, stream>>>(static_cast<const _Float16*>(table), width, batch, indices,
                                     static_cast<const int*>(nullptr), num_hots,
                                     static_cast<const _Float16*>(nullptr), false,
                                     static_cast<_Float16*>(out), SLICES, /*stream_rows=*/false);
}

extern "C" int variant_count() { return 13; }
extern "C" const char* variant_name(int id) {
  static const char* names[] = {"u8 plain lb1024", "u8 plain lb256", "u4 plain lb256",
                                "u16 plain lb256", "u4 pipelined lb256", "u8 pipelined lb256",
                                "u8 pipelined lb512", "u2 pipelined lb256", "u6 plain lb256",
                                "u12 plain lb256", "u8 slices2 lb1024", "u8 slices4 lb1024",
                                "u8 slices8 lb1024"};
  return names[id];
}
extern "C" void variant_launch(int id, const void* table, int width, int batch, const int* indices,
                               int num_hots, void* out, int samples_per_block, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (id) {
    case 0: Launch<8, false, 1024>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 1: Launch<8, false, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 2: Launch<4, false, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 3: Launch<16, false, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 4: Launch<4, true, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 5: Launch<8, true, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 6: Launch<8, true, 512>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 7: Launch<2, true, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 8: Launch<6, false, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 9: Launch<12, false, 256>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 10: Launch<8, false, 1024, 2>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 11: Launch<8, false, 1024, 4>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
    case 12: Launch<8, false, 1024, 8>(table, width, batch, indices, num_hots, out, samples_per_block, s); break;
  }
}
