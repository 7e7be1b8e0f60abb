// Store discipline of the small row kernels (bank, allocation tables, decode records, anchor projection).
//
// Finding of round 2 ("the eager two-stream fault"; data under profiles/r02_bank_get_fault/):
// bank_get_kernel, as hipcc scheduled it, issued the store of its pass-through columns while seven loads were
// still outstanding and then consumed the load results behind COUNTED waits (s_waitcnt vmcnt(5)/(3)/(2)/(1), which
// rely on that store retiring in issue order with the loads). Beside a busy second hardware queue (eager backbone
// convolutions) about 1 launch in 100 then evaluated one multiply-add of lanes 48-63 of ONE wave with an operand
// read as zero (the last quarter-wave pass; the registers themselves held the right values before and after: the
// kernel's own self-check build logged them). The same arithmetic with every load retired before the first store
// (0 faults in 100 repetitions against 17 for the original, same box, same session) does not fault.
// Rule: results into registers, every load retired, THEN the stores; tools/isa_store_scan.py (a CPU test) checks the
// generated ISA for stores issued among outstanding loads in front of a counted wait.
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

}  // namespace simpb
