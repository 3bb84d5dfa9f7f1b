// What a process that has touched the GPU costs at its END, whatever it did: hipcc -O2 -o /tmp/exit_cost exit_cost.hip ;
// /tmp/exit_cost [MB of device memory to hold, default 1]   -- prints the seconds from start to the last line of main;
// run it under `time` (or tools/exit_cost.py): the difference is the driver taking the process's queues and address
// space down.  (The host program's 0.13-0.24 s "outside every phase", DESIGN s7.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv)
{
    const auto t0 = std::chrono::steady_clock::now();
    const size_t mb = argc > 1 ? strtoul(argv[1], nullptr, 10) : 1;
    int *p = nullptr;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&p, mb << 20) != hipSuccess)
        return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p);
    if (hipDeviceSynchronize() != hipSuccess)
        return 1;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%.4f s inside main (device start, %zu MB, one kernel)\n", s, mb);
    fflush(stdout);
    if (argc > 2)
        _exit(0);        // any second argument: leave without the runtime's exit handlers
    return 0;
}
