// lds_dma.h - LDS-DMA through a buffer descriptor (device code only; included by the .hip files).
//
// buffer_load_dwordx4 ... offen lds: 16 B per lane from (descriptor base + soff + voff) straight into LDS at
// (wave-uniform l) + lane*16, no VGPR round trip and no per-lane 64-bit address math.  Lanes whose offset fails the
// descriptor's range check write ZEROS (verified by scripts/micro/buflds_oob.hip, which also shows soff takes part in
// the check): zero padding rides on that - a padded lane carries PAD_OFF, beyond every descriptor used here.
#pragma once

typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned PAD_OFF = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int num_bytes = 0x7ffffff0) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, num_bytes, 0x00020000);
}
__device__ __forceinline__ void blds16(rsrc_t r, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

// Hazard hunting (-DTSD_JITTER builds only, never shipped): a random wave-level delay at the points where waves meet or part.  A kernel
// whose waves are correctly ordered gives the same bits whatever the delays; a hazard that a fixed schedule hides becomes a run-to-run
// difference within a few launches (scripts/diag_race3.py counts distinct results).
__device__ __forceinline__ void tsd_jitter() {
#ifdef TSD_JITTER
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
  if (((t >> 2) & 3) == 0) {
    const int n = (int)((t >> 4) & 31);
    for (int z = 0; z < n; z++) __builtin_amdgcn_s_sleep(4);
  }
#endif
}
