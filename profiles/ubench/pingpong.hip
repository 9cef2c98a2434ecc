// Round 6 micro-benchmark: what a hand-off between two work-groups on different compute units costs on MI355X -- the price of spreading one channel's window of the
// closed-loop kernel over cooperating work-groups.  Two work-groups pass a tagged 64-bit word back and forth (value | epoch << 32) with relaxed
// AGENT-scope atomics, no fences: A writes word e, B polls until it sees tag e and answers, A polls for the answer.  Reported: nanoseconds per ONE-WAY hand-off
// (round trip / 2) for partner blocks on the same XCD (block ids 8 apart) and on different XCDs (1 apart), and the same with 8 words per hand-off (a wave's lanes).
// hipcc --offload-arch=gfx950 -O3 -o profiles/ubench/pingpong profiles/ubench/pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void pingpong(unsigned long long* box, int partner_stride, int rounds, int words, long long* clocks)
{
    // blocks 0 and partner_stride play; everybody else leaves
    const int b = blockIdx.x;
    if (b != 0 && b != partner_stride) return;
    const int lane = threadIdx.x;
    if (lane >= words) return;
    unsigned long long* mine = box + (b == 0 ? 0 : 64) + lane;      // what I write
    unsigned long long* theirs = box + (b == 0 ? 64 : 0) + lane;    // what I read
    const long long t0 = wall_clock64();
    for (int e = 1; e <= rounds; e++)
        {
            if (b == 0) __hip_atomic_store(mine, (static_cast<unsigned long long>(e) << 32) | static_cast<unsigned>(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long v;
            long long spins = 0;
            do
                {
                    v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > 20000000) return;  // never hang the box
                }
            while (__any((v >> 32) != static_cast<unsigned long long>(e)));
            if (b != 0) __hip_atomic_store(mine, (static_cast<unsigned long long>(e) << 32) | static_cast<unsigned>(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (b == 0 && lane == 0) clocks[0] = wall_clock64() - t0;
}

int main()
{
    unsigned long long* box;
    long long* clocks;
    hipMalloc(&box, 128 * sizeof(unsigned long long));
    hipMalloc(&clocks, sizeof(long long));
    const int rounds = 20000;
    for (int words : {1, 8, 32})
        for (int stride : {8, 1, 3, 16})
            {
                hipMemset(box, 0, 128 * sizeof(unsigned long long));
                hipMemset(clocks, 0, sizeof(long long));
                hipLaunchKernelGGL(pingpong, dim3(stride + 1), dim3(64), 0, 0, box, stride, rounds, words, clocks);
                if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
                long long c = 0;
                hipMemcpy(&c, clocks, sizeof c, hipMemcpyDeviceToHost);
                // wall_clock64 ticks at 100 MHz on gfx9
                std::printf("%2d word(s), partner %2d blocks away (%s): %.0f ns per one-way hand-off\n", words, stride, (stride % 8 == 0) ? "same XCD" : "other XCD",
                    c * 10.0 / rounds / 2.0);
            }
    return 0;
}
