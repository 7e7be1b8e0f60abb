// Store discipline of the small row kernels (bank, allocation tables, decode records, anchor projection).
//
// Round 2 ("the two-stream fault"): while a kernel built on v_mfma_f32_32x32x16_f16 ran on another
// stream, vector arithmetic of these kernels went wrong in lanes 48-63 of a wave now and then. The cause was that
// instruction (csrc/mfma_f16.h no longer issues it); the kernels that were hit first and most (bank_get: 1 launch in 100,
// dfa_points) were the ones whose schedule had a vector store in flight while COUNTED waits (s_waitcnt vmcnt(N > 0))
// released the consumers of load results, and the same arithmetic with every load retired before it ran clean in the
// same session (profiles/r02_bank_get_fault/). The row kernels therefore keep this shape: operands into registers, every
// load retired, arithmetic, results pinned, every load retired, stores. Hardening, checked on the generated ISA by
// tools/isa_store_scan.py (a CPU test); it costs nothing measurable in kernels of a few microseconds.
#pragma once
#include <hip/hip_runtime.h>

namespace simpb {

// Pins `v` as computed at this point (the optimiser cannot sink its producers below, nor hoist a store above).
template <class T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)::"memory"); }

// Every vector-memory load of this wave has written its registers; no store may be scheduled above this line.
__device__ __forceinline__ void loads_retired() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Every vector-memory operation of this wave (stores included: gfx9 counts them in vmcnt) has completed. At the end of a
// loop body that stores: the next trip's loads then never wait on a counted vmcnt with this trip's store in flight.
__device__ __forceinline__ void stores_retired() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace simpb
