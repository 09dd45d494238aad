// How should 64 MiB of PAGEABLE host memory (a caller's Vec<C>) cross PCIe when work is waiting for its ranges?  (round 5, h2_msm from host
// slices.)  hipcc -O2 bench/ubench_h2d.hip -o build/ubench_h2d -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
    const size_t bytes = (size_t)64 << 20;
    std::vector<char> host(bytes, 1);
    void *d = nullptr;
    unsigned *d_ctr = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc((void **)&d_ctr, 64));
    hipStream_t copy, work;
    CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) CK(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice));
    auto best_of = [&](auto fn) { double b = 1e30; for (int r = 0; r < 5; ++r) { double t0 = now_ms(); fn(); b = std::min(b, now_ms() - t0); } return b; };
    printf("one hipMemcpy (pageable)                      : %.3f ms\n", best_of([&] { CK(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice)); }));
    for (int Q : {2, 4, 8, 16, 32}) {
        const double t = best_of([&] {
            for (int q = 0; q < Q; ++q) CK(hipMemcpyAsync((char *)d + bytes / Q * q, host.data() + bytes / Q * q, bytes / Q, hipMemcpyHostToDevice, copy));
            CK(hipStreamSynchronize(copy));
        });
        printf("%2d x hipMemcpyAsync on a stream (pageable)     : %.3f ms\n", Q, t);
    }
    {   // how long does the CALL hold the host?
        double t0 = now_ms();
        CK(hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, copy));
        const double call = now_ms() - t0;
        CK(hipStreamSynchronize(copy));
        printf("hipMemcpyAsync(64 MiB pageable): the call returns after %.3f ms, done after %.3f ms\n", call, now_ms() - t0);
    }
    {
        double t0 = now_ms();
        CK(hipHostRegister(host.data(), bytes, hipHostRegisterDefault));
        const double reg = now_ms() - t0;
        t0 = now_ms();
        CK(hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, copy));
        const double call = now_ms() - t0;
        CK(hipStreamSynchronize(copy));
        const double cp = now_ms() - t0;
        t0 = now_ms();
        CK(hipHostUnregister(host.data()));
        printf("hipHostRegister %.3f ms, async copy of registered memory: call %.3f ms, done %.3f ms, unregister %.3f ms\n", reg, call, cp, now_ms() - t0);
    }
    // kernel launches from this thread while another thread is inside a pageable copy
    for (int Q : {1, 8}) {
        std::atomic<int> go{0};
        double copy_ms = 0;
        std::thread th([&] {
            while (!go.load()) {}
            double t0 = now_ms();
            for (int q = 0; q < Q; ++q) CK(hipMemcpyAsync((char *)d + bytes / Q * q, host.data() + bytes / Q * q, bytes / Q, hipMemcpyHostToDevice, copy));
            CK(hipStreamSynchronize(copy));
            copy_ms = now_ms() - t0;
        });
        CK(hipDeviceSynchronize());
        go.store(1);
        double t0 = now_ms();
        const int L = 200;
        for (int i = 0; i < L; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, work, d_ctr);
        const double enq = now_ms() - t0;
        CK(hipStreamSynchronize(work));
        const double done = now_ms() - t0;
        th.join();
        printf("200 launches while a helper thread copies 64 MiB in %d piece(s): enqueued in %.3f ms (%.1f us each), executed by %.3f ms; the copy took %.3f ms\n", Q, enq,
               enq * 1e3 / L, done, copy_ms);
    }
    {
        CK(hipDeviceSynchronize());
        double t0 = now_ms();
        const int L = 200;
        for (int i = 0; i < L; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, work, d_ctr);
        const double enq = now_ms() - t0;
        CK(hipStreamSynchronize(work));
        printf("200 launches, nothing else running: enqueued in %.3f ms (%.1f us each), executed by %.3f ms\n", enq, enq * 1e3 / L, now_ms() - t0);
    }
    return 0;
}
