// Can the CPU write device memory directly (fine-grained device allocation through the PCIe BAR), and how long until a
// resident kernel sees it -- against the kernel pulling the same word from pinned host memory?  (tools/micro)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_echo(const unsigned long long *in, unsigned long long *out_host, int rounds, int sys_scope)
{
    unsigned long long last = 0;
    for (int r = 0; r < rounds; ++r)
    {
        unsigned long long v;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;)
        {
            v = sys_scope ? __hip_atomic_load(in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
                          : __hip_atomic_load(in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (v != last) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return;
        }
        last = v;
        __hip_atomic_store(out_host, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main()
{
    unsigned long long *h_out, *h_in, *d_in = nullptr;
    CK(hipHostMalloc((void **)&h_out, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_in, 64, hipHostMallocDefault));
    hipError_t e = hipExtMallocWithFlags((void **)&d_in, 64, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s ptr %p\n", hipGetErrorString(e), (void *)d_in);
    for (int mode = 0; mode < 2; ++mode)
    {
        unsigned long long *in = mode == 0 ? h_in : d_in;
        if (!in) continue;
        if (mode == 1)
        { // is it host-writable at all?
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, in) == hipSuccess) printf("  attr: type %d host %p dev %p managed %d\n", (int)a.type, a.hostPointer, a.devicePointer, a.isManaged);
            CK(hipMemset(in, 0, 64));
            CK(hipDeviceSynchronize());
        }
        *h_out = 0;
        if (mode == 0) *h_in = 0;
        const int rounds = 2000;
        hipLaunchKernelGGL(k_echo, dim3(1), dim3(64), 0, 0, in, h_out, rounds, mode == 0 ? 1 : 1);
        double tot = 0;
        for (int r = 1; r <= rounds; ++r)
        {
            auto t0 = std::chrono::steady_clock::now();
            *(volatile unsigned long long *)in = (unsigned long long)r; // CPU store: to pinned host memory, or through the BAR
            __builtin_ia32_sfence(); // device memory is mapped write-combining: without the fence the store waits in the WC buffer
            while (*(volatile unsigned long long *)h_out != (unsigned long long)r) {}
            tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        CK(hipDeviceSynchronize());
        printf("%s: round trip %.2f us (host store -> kernel sees it -> kernel's store to pinned host -> host sees it)\n",
               mode == 0 ? "pull from pinned host memory" : "push into fine-grained device memory", tot / rounds * 1e6);
    }
    return 0;
}
