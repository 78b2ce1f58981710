// dl_scalar_prefetch.h -- warm the scalar cache with the kernel arguments (or another block of uniform data) at kernel entry.
//
// Kernel arguments are cold at every launch (the runtime writes them to a fresh slot of its ring) and are read field by field through the scalar cache; the compiler
// places each s_load next to its first use, so every 64-byte line of the arguments is a scalar-cache miss ON the dependent chain of a latency-bound kernel's prologue
// (theory kernel of the headline: 2 KB description, 11.9 -> 11.05 us with the lines requested at entry).
#pragma once
#include <hip/hip_runtime.h>

// Requesting the lines at entry (results discarded: the loads share their destination registers) turns the misses into one round trip; the real loads then hit
// the scalar cache.
// 64-byte lines [A0, A1) and [B0, B1) (byte offsets from p).  ONE asm statement: the destination registers belong to it from the first request to the wait.
template <int A0, int A1, int B0, int B1>
__device__ __forceinline__ void dl_scalar_prefetch(const void* p) {
    __asm__ volatile(
        "s_mov_b32 s83, %1\n"
        "1: s_load_dwordx16 s[84:99], %0, s83\n\ts_add_u32 s83, s83, 64\n\ts_cmp_lt_u32 s83, %2\n\ts_cbranch_scc1 1b\n"
        "s_mov_b32 s83, %3\n"
        "2: s_load_dwordx16 s[84:99], %0, s83\n\ts_add_u32 s83, s83, 64\n\ts_cmp_lt_u32 s83, %4\n\ts_cbranch_scc1 2b\n"
        "s_waitcnt lgkmcnt(0)"
        : : "s"(p), "n"(A0), "n"(A1), "n"(B0), "n"(B1)
        : "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "scc", "memory");
}

// the first BYTES bytes of the kernel arguments, asked for (and waited for) by the first wavefront of the workgroup: the others' loads of the same lines ride on
// the requests in flight (every wavefront asking is slower than nobody asking: the scalar cache serves several CUs)
template <int BYTES>
__device__ __forceinline__ void dl_kernarg_prefetch() {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 64) return;
    constexpr int END = (BYTES + 63) / 64 * 64, MID = END > 64 ? 64 : 0;   // (two ranges in the statement: first line, the rest)
    dl_scalar_prefetch<0, 64, MID == 0 ? 0 : 64, MID == 0 ? 64 : END>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
}
