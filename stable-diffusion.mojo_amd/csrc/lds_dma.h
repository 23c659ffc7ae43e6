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

// Counted waits carry their meaning into the assembly (round 5).  `s_waitcnt vmcnt(N)` with N > 0 is only correct if the N youngest
// vector-memory instructions at that point are exactly what the SOURCE thinks they are - and hipcc is free to merge, split or move
// plain loads (round 4 shipped a vmcnt(15 + 25) where the compiler had issued 17 loads, not 25).  Every source-written counted wait
// therefore states its accounting in a comment behind the instruction:
//     s_waitcnt vmcnt(N) ; tsd-wait dma=<D> other=<E> ppt=<P>,<P2>
//   D   LDS-DMA instructions (pieces of YOUNGER tiles) that may stay in flight
//   E   other vector-memory loads counted on top (16-byte loads only: the compiler cannot merge those), N = D + E
//   P   LDS-DMA instructions this wave issues per tile (P2: a second tile size of the same stream, 0 = none)
// tools/isa_lint.py reads the compiler's assembly of the shipped objects (`make` keeps it: -save-temps=obj), rebuilds each kernel's
// control-flow graph and checks on EVERY path into the wait that the N youngest vector-memory instructions hold at least E plain
// loads and no store (otherwise older DMA pieces than intended stay in flight), that whole tiles are issued between two waits, and
// that the stream / register / scratch figures equal the committed table (csrc/isa_contract.json).  __graft_entry__.build() and the
// CPU test-suite run it: a compiler that changes what a counted wait counts fails the BUILD, not one run in a hundred.
template <int DMA, int OTHER = 0, int PPT = 0, int PPT2 = 0>
__device__ __forceinline__ void wait_vm_counted() {
  static_assert(DMA >= 0 && OTHER >= 0 && DMA + OTHER < 64, "vmcnt is a 6-bit counter");
  static_assert(DMA == 0 || PPT > 0, "a wait that leaves DMA pieces in flight names the pieces per tile");
  asm volatile("s_waitcnt vmcnt(%0) ; tsd-wait dma=%1 other=%2 ppt=%3,%4" ::"n"(DMA + OTHER), "n"(DMA), "n"(OTHER), "n"(PPT), "n"(PPT2) : "memory");
}

// Brackets around an if / else chain of counted waits of which exactly one executes (see wait_ring in kernels_gemm.hip)
__device__ __forceinline__ void wait_alt_begin() { asm volatile("; tsd-wait-alt begin" ::: "memory"); }
__device__ __forceinline__ void wait_alt_end() { asm volatile("; tsd-wait-alt end" ::: "memory"); }

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
