// LDS-DMA helpers shared by the MFMA kernels (gfx950).
#pragma once
#include "common.hpp"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;
static __device__ uint4 g_zero_page[16];   // 256 B of zeros: DMA source for padding rows (one copy per translation unit)

// Counted wait: at most N of this wave's vector-memory operations (LDS-DMA pieces included) may still be in flight.
// -DMI355_DMA_DRAIN (make drain -> libmi355conv_drain.so) turns EVERY counted wait into vmcnt(0): the conservative schedule the
// hand-counted ones are A/B-ed against (tests/test_gpu_conv.py::test_counted_vmcnt_matches_drained_build compares the two builds
// bit for bit — a wrong count shows up as a difference, not as a hang).
#ifdef MI355_DMA_DRAIN
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#elif defined(MI355_T_NOWAIT)      // timing-only build (stale operands): what the counted waits cost
template <int N> __device__ __forceinline__ void wait_vmcnt() { if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
#endif

// One 1-KiB LDS-DMA piece: lane i's 16 bytes at `gsrc` land at LDS byte address lds_base + 16*i.
// Issued from inline asm ON PURPOSE: hipcc's waitcnt pass then does not know an LDS write is pending
// and does not drain vmcnt(0) in front of the next ds_read (which would serialise the whole ring);
// ordering is ours: counted s_waitcnt vmcnt + s_barrier before a stage is read (cdna guide 5.7).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
// The same piece with a wave-uniform base (an SGPR pair) and a 32-bit per-lane byte offset: one VGPR per source instead of a
// 64-bit address pair — what the 512-thread kernels, which have no registers to spare, keep per piece.
__device__ __forceinline__ void dma16_sv(const void* sbase, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}
// ... and with M0 declared clobbered instead of saved and restored (two scalar instructions per piece less; for kernels in which
// nothing else lives in M0 — no s_movrel, no ds_*_gs / GWS), the LDS destination given as a byte address (an integer: casting
// a generic pointer to LDS costs a null check per piece)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "clobber list contains reserved registers: m0" — that is the point
__device__ __forceinline__ void dma16_sv_m0(const void* sbase, unsigned voff, unsigned lds_byte) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_byte)
      : "memory", "m0");
}
#pragma clang diagnostic pop
// LDS-DMA through a buffer descriptor: source = descriptor base + scalar offset `soff` + the lane's 32-bit byte offset, and the
// hardware's range check does the padding — a lane whose offset is >= the descriptor's num_records reads ZEROS (0x80000000 with
// the num_records below; the scalar offset takes no part in the check), and a descriptor with num_records = 0 zero-fills the
// whole piece.  No zero page, no per-lane address select, no EXEC masking, and every wave issues the same number of pieces
// whatever its lanes' validity (the counted vmcnt waits rely on that).  M0 clobbered as above.
typedef int __attribute__((ext_vector_type(4))) bufdesc_t;
constexpr unsigned DMA_PAD = 0x80000000u;            // a lane offset that is always out of range
__device__ __forceinline__ bufdesc_t make_buf(const void* base, bool valid = true) {
  const unsigned long long p = (unsigned long long)base;
  bufdesc_t d;
  d[0] = (int)(unsigned)p;
  d[1] = (int)(unsigned)((p >> 32) & 0xffffu);       // stride 0: a raw buffer
  d[2] = valid ? (int)DMA_PAD : 0;                    // num_records (bytes): every real offset is < 2^31
  d[3] = 0x00020000;                                  // DATA_FORMAT = 32 (untyped dword loads ignore the format fields)
  return d;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_buf(bufdesc_t desc, unsigned voff, unsigned soff, unsigned lds_byte) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %2 offen lds"
      :
      : "v"(voff), "s"(desc), "s"(soff), "s"(lds_byte)
      : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)p);
}

