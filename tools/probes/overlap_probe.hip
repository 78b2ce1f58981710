// Can a dependent kernel be dispatched while its producer still runs?  Stream S: kernel A (every workgroup lives ~8 us; its last workgroup raises a flag when it
// starts).  Stream T: hipStreamWaitValue64(flag >= epoch), then kernel B whose workgroups need the whole LDS of a CU (they become resident as A's leave).
// Prints, per trial: start of the first / last B workgroup and end of B relative to the start of A, against the serial arrangement (B after A on one stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void kernel_a(unsigned long long* flag, unsigned long long epoch, unsigned long long* stamps, int life_ticks) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        if (blockIdx.x == gridDim.x - 1 && flag) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    lds[threadIdx.x] = (double)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)life_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(1024) void kernel_b(unsigned long long* stamps, int life_ticks) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = t0;
    lds[threadIdx.x] = 1.;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)life_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    hipStream_t S, T;
    CHECK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&T, hipStreamNonBlocking));
    unsigned long long *flag = nullptr, *sa, *sb;
    hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal memory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    CHECK(hipMemset(flag, 0, 8));
    const int NA = 1024, NB = 256;
    CHECK(hipMalloc(&sa, NA * 16)); CHECK(hipMalloc(&sb, NB * 16));
    CHECK(hipFuncSetAttribute((const void*)kernel_b, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<unsigned long long> ha(2 * NA), hb(2 * NB);
    for (int mode = 0; mode < 2; ++mode) {
        for (int trial = 0; trial < 6; ++trial) {
            const unsigned long long epoch = 1 + mode * 100 + trial;
            if (mode == 0) {   // serial: B after A in one stream
                hipLaunchKernelGGL(kernel_a, dim3(NA), dim3(256), 36 * 1024, S, nullptr, epoch, sa, 800);
                hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(1024), 144 * 1024, S, sb, 500);
            } else {           // B on its own stream, released by the last A workgroup's flag; S then waits for B
                e = hipStreamWaitValue64(T, flag, epoch, hipStreamWaitValueGte, 0xffffffffffffffffull);
                if (e != hipSuccess) { printf("hipStreamWaitValue64: %s\n", hipGetErrorString(e)); return 1; }
                hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(1024), 144 * 1024, T, sb, 500);
                CHECK(hipEventRecord(ev, T));
                hipLaunchKernelGGL(kernel_a, dim3(NA), dim3(256), 36 * 1024, S, flag, epoch, sa, 800);
                CHECK(hipStreamWaitEvent(S, ev, 0));
            }
            CHECK(hipStreamSynchronize(S)); CHECK(hipStreamSynchronize(T));
            CHECK(hipMemcpy(ha.data(), sa, NA * 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), sb, NB * 16, hipMemcpyDeviceToHost));
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b0max = 0, b1 = 0;
            for (int i = 0; i < NA; ++i) { a0 = std::min(a0, ha[2 * i]); a1 = std::max(a1, ha[2 * i + 1]); }
            for (int i = 0; i < NB; ++i) { b0 = std::min(b0, hb[2 * i]); b0max = std::max(b0max, hb[2 * i]); b1 = std::max(b1, hb[2 * i + 1]); }
            printf("%s trial %d: A ends %.2f us; B first start %.2f, last start %.2f, ends %.2f us (after the start of A)\n", mode ? "overlap" : "serial ", trial,
                   0.01 * (a1 - a0), 0.01 * ((double)b0 - (double)a0), 0.01 * ((double)b0max - (double)a0), 0.01 * ((double)b1 - (double)a0));
        }
    }
    // steady state: N iterations of A -> B (B needs A), wall time per iteration
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            const int N = 200;
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, S));
            for (int it = 0; it < N; ++it) {
                const unsigned long long epoch = 1000 + mode * 100000 + rep * 1000 + it;
                if (mode == 0) {
                    hipLaunchKernelGGL(kernel_a, dim3(NA), dim3(256), 36 * 1024, S, nullptr, epoch, sa, 800);
                    hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(1024), 144 * 1024, S, sb, 500);
                } else {
                    CHECK(hipStreamWaitValue64(T, flag, epoch, hipStreamWaitValueGte, 0xffffffffffffffffull));
                    hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(1024), 144 * 1024, T, sb, 500);
                    CHECK(hipEventRecord(ev, T));
                    hipLaunchKernelGGL(kernel_a, dim3(NA), dim3(256), 36 * 1024, S, flag, epoch, sa, 800);
                    CHECK(hipStreamWaitEvent(S, ev, 0));
                }
            }
            CHECK(hipEventRecord(e1, S));
            CHECK(hipStreamSynchronize(S)); CHECK(hipStreamSynchronize(T));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s steady state: %.2f us per iteration (A 8 us + B 5 us of pure workgroup life)\n", mode ? "overlap" : "serial ", 1e3 * ms / N);
        }
    }
    return 0;
}
