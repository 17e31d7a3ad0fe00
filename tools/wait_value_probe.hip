// Probe: can a stream wait on a host-written word (hipStreamWaitValue32), so that the NEXT launch of a host-stepped chain is
// queued before its observations exist and released by a plain store?  Prints what the runtime accepts and two latencies:
// release -> completion word of a pre-queued empty kernel, and hipLaunch -> completion word of the same kernel launched late.
//   hipcc --offload-arch=gfx950 -O2 tools/wait_value_probe.hip -o /tmp/wvp && /tmp/wvp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <thread>

__global__ void tiny(int *p) { if (threadIdx.x == 0 && p) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue %d\n", can);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    volatile uint32_t *done = nullptr;
    hipHostMalloc((void **)&done, 64, hipHostMallocDefault);
    *done = 0;
    // candidates for the gate word: page-locked host memory, signal memory
    uint32_t *gate_host = nullptr;
    hipHostMalloc((void **)&gate_host, 64, hipHostMallocDefault);
    *gate_host = 0;
    void *gate_sig = nullptr;
    hipError_t e = hipExtMallocWithFlags(&gate_sig, 8, hipMallocSignalMemory);
    printf("signal memory alloc: %s ptr %p\n", hipGetErrorName(e), gate_sig);
    hipPointerAttribute_t at{};
    if (gate_sig && hipPointerGetAttributes(&at, gate_sig) == hipSuccess)
        printf("signal memory: type %d hostPointer %p devicePointer %p isManaged %d\n", (int)at.type, at.hostPointer, at.devicePointer, at.isManaged);
    (void)hipGetLastError();
    for (int which = 0; which < 2; ++which) {
        volatile uint32_t *gate = which == 0 ? gate_host : (volatile uint32_t *)(at.hostPointer ? at.hostPointer : nullptr);
        void *gate_dev = which == 0 ? (void *)gate_host : gate_sig;
        if (!gate || !gate_dev) { printf("candidate %d: no host-visible pointer\n", which); continue; }
        *gate = 0;
        e = hipStreamWaitValue32(s, gate_dev, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        printf("candidate %d (%s): hipStreamWaitValue32 -> %s\n", which, which == 0 ? "hipHostMalloc" : "signal memory", hipGetErrorName(e));
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        *gate = 1;   // release the probe wait
        hipStreamSynchronize(s);
        double rel = 0, late = 0;
        const int N = 200;
        uint32_t seq = 0, gv = 1;
        for (int i = 0; i < N; ++i) {
            ++gv; ++seq;
            hipStreamWaitValue32(s, gate_dev, gv, hipStreamWaitValueGte, 0xFFFFFFFFu);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, (int *)nullptr);
            hipStreamWriteValue32(s, (void *)done, seq, 0);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            if (*done == seq) { printf("the wait did not hold\n"); break; }
            const double t0 = now_us();
            *gate = gv;
            while (*done != seq) {}
            rel += now_us() - t0;
        }
        for (int i = 0; i < N; ++i) {
            ++seq;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            const double t0 = now_us();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, (int *)nullptr);
            hipStreamWriteValue32(s, (void *)done, seq, 0);
            while (*done != seq) {}
            late += now_us() - t0;
        }
        printf("candidate %d: release -> done %.1f us; launch -> done %.1f us (empty kernel, idle stream)\n", which, rel / N, late / N);
    }
    return 0;
}
